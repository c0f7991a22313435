#!/bin/bash
# Generic PMC pass: profiles/pmc_cmd.sh <tag> "<counters...>" <command...>
# Own run per counter group, --kernel-trace only alongside --pmc (MI355X_MICROARCH.md; gpurun refuses other mixes).
# Output: gpurun_out/pmc_<tag>.txt = per-kernel sums / means of every counter in the group.
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc_$tag
(cd $R && timeout ${PP_TIMEOUT:-180} rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/pmc_$tag -o r -- "$@" > $R/gpurun_out/pmc_$tag.log 2>&1)
db=$(find /tmp/pmc_$tag -name '*.db' | head -1)
python - "$db" "${PMC_FILTER:-k_}" > $R/gpurun_out/pmc_$tag.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, counter_name, count(*), avg(counter_value), avg(duration)/1e3 from pmc_events group by name, counter_name order by name").fetchall()
by = {}
for n, cn, k, v, d in rows:
    by.setdefault(n, {"launches": k, "avg_us": d})[cn] = v
for n, d in sorted(by.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"]):
    if sys.argv[2] not in n: continue
    extra = "  ".join(f"{k} {v:.4g}" for k, v in d.items() if k not in ("launches", "avg_us"))
    print(f"{n[:84]:84s} n {d['launches']:5d} avg_us {d['avg_us']:9.1f}  {extra}")
PY
echo "== $tag"; head -${PMC_HEAD:-12} $R/gpurun_out/pmc_$tag.txt | cut -c1-400
