#!/usr/bin/env python
"""Per-kernel means of rocprofv3 PMC counters from a results .db (pmc_events table), one line per kernel and counter.
Launches of the non-gated streaming GEMM are split by duration into the MoE down projection (> 150 us at S = 552) and the QKV / O
projections, which share one instantiation.

  python profiles/pmc_table.py results.db COUNTER [COUNTER ...] [--title "..."]"""
import sqlite3
import sys

args = [a for a in sys.argv[1:]]
title = ""
if "--title" in args:
    i = args.index("--title")
    title = args[i + 1]
    del args[i:i + 2]
db, counters = args[0], args[1:]
c = sqlite3.connect(db)
if title:
    print("# " + title)
print("# kernel | launch records | avg_us | " + " | ".join(counters) + "   (per-launch means)")
rows = {}
for cn in counters:
    for name, val, dur in c.execute("select name, counter_value, duration from pmc_events where counter_name = ?", (cn,)):
        key = name[:90]
        if "k_gemm_sp<false" in name or "k_gemm_ps<false" in name:
            key += "  [MoE down]" if dur / 1e3 > 150 else "  [QKV / O / encoder Linears]"
        d = rows.setdefault(key, {})
        e = d.setdefault(cn, [0, 0.0, 0.0])
        e[0] += 1; e[1] += val; e[2] += dur / 1e3
order = sorted(rows, key=lambda k: -max(v[2] for v in rows[k].values()))
for k in order[:24]:
    d = rows[k]
    n = max(v[0] for v in d.values())
    us = max(v[2] / v[0] for v in d.values())
    print(f"{k}\t{n}\t{us:.2f}\t" + "\t".join(f"{d[cn][1] / d[cn][0]:.1f}" if cn in d else "-" for cn in counters))
