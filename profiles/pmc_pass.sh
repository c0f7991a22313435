#!/bin/bash
# HBM-traffic PMC pass (own run, --kernel-trace only alongside --pmc, as MI355X_MICROARCH.md prescribes):
#   FETCH_SIZE (TCC, 3 slots) and WRITE_SIZE (2 slots) cannot share a pass -> two runs.
# Output: gpurun_out/pmc_<counter>.txt = per-kernel mean counter value per launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ctr in ${PMC_CTRS:-FETCH_SIZE WRITE_SIZE}; do
  rm -rf /tmp/pmc_$ctr
  timeout ${PP_TIMEOUT:-120} rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$ctr -o r -- python $R/bench.py --layers ${PP_LAYERS:-4} --steps 8 --warmup 2 --no-cpu-baseline $PMC_EXTRA > $R/gpurun_out/pmc_$ctr.log 2>&1
  db=$(find /tmp/pmc_$ctr -name '*.db' | head -1)
  cp $db $R/gpurun_out/pmc_$ctr.db
  python - "$db" $ctr > $R/gpurun_out/pmc_$ctr.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(pmc_events)").fetchall()]
print("# columns:", cols)
q = """select name, count(*), avg(counter_value), min(counter_value), max(counter_value), avg(duration)/1e3
       from pmc_events where counter_name = ? group by name order by 3 desc"""
try:
    rows = c.execute(q, (sys.argv[2],)).fetchall()
except Exception as e:
    print("# query failed:", e)
    rows = []
print(f"# counter {sys.argv[2]}: name, launches, mean, min, max, avg_us")
for r in rows[:40]:
    print(f"{r[0][:90]}\t{r[1]}\t{r[2]:.1f}\t{r[3]:.1f}\t{r[4]:.1f}\t{r[5]:.2f}")
PY
  echo "== $ctr"; head -20 $R/gpurun_out/pmc_$ctr.txt | cut -c1-200
done
