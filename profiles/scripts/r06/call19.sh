#!/bin/bash
# r06 call 19: gate|up block count at TP = 1 (7168 row groups: 384 blocks = 18.67 rounds; 399 = 18, 448 = 16, 512 = 14, 256 = 28 whole rounds), two interleaved rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
T0=$(date +%s)
B="python3 bench.py --no-cpu-baseline --phase-iters 1 --phase-warmup 1 --steps 64 --warmup 8"
for rep in 1 2; do
  for g in 0 256 399 448 512 768; do
    timeout 300 $B --tune dec_gateup_grid=$g > $O/c19_grid${g}_$rep.json 2>> $O/c19.err
  done
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c19_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "ms", d["ms_per_step"], "gateup live us", d["roofline"]["avg_launch_us"])
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
echo "total $(( $(date +%s) - T0 )) s"
