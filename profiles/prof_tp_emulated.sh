#!/bin/bash
# per-kernel decode times of ONE rank's shard at TP=N with the collective skipped (bench.py --emulate-tp N)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in "$@"; do
  rm -rf /tmp/ptp_$n
  timeout 120 rocprofv3 --kernel-trace -d /tmp/ptp_$n -o r -- python $R/bench.py --layers 8 --steps 24 --warmup 4 --no-cpu-baseline --emulate-tp $n > $R/gpurun_out/ptp_$n.log 2>&1
  echo "== emulated TP=$n (rank compute only)"
  python $R/profiles/summarize.py $(find /tmp/ptp_$n -name '*.db' | head -1) | grep -E "k_dec" | cut -c1-120 | tee $R/gpurun_out/ptp_$n.txt
done
