#!/bin/bash
# SQ counter pass (own run): where do the waves of a kernel spend their cycles, LDS bank conflicts.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc_sq
timeout ${PP_TIMEOUT:-90} rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace -d /tmp/pmc_sq -o r -- python $R/bench.py --layers 2 --steps 4 --warmup 2 --no-cpu-baseline $1 > $R/gpurun_out/pmc_sq.log 2>&1
db=$(find /tmp/pmc_sq -name '*.db' | head -1)
python - "$db" > $R/gpurun_out/pmc_sq.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, counter_name, count(*), avg(counter_value), avg(duration)/1e3 from pmc_events group by name, counter_name order by name").fetchall()
cur = None
for n, cn, k, v, d in rows:
    if "k_gemm" not in n and "k_attn" not in n and "k_dec_gateup" not in n: continue
    if n != cur:
        print(f"== {n[:80]}  launches {k} avg_us {d:.1f}"); cur = n
    print(f"   {cn:24s} {v:16.0f}")
PY
cat $R/gpurun_out/pmc_sq.txt
