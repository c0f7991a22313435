#!/bin/bash
# Per-kernel timing of decode-kernel variants on a reduced-depth model (kernel durations do not depend on
# depth): rocprofv3 kernel trace per variant -> gpurun_out/var_<name>.txt.  usage: prof_variants.sh name=tune ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%%:*}; tune=${spec#*:}
  rm -rf /tmp/pv_$name
  timeout 300 rocprofv3 --kernel-trace -d /tmp/pv_$name -o r -- python $R/bench.py --layers 8 --steps 24 --warmup 4 --no-cpu-baseline ${tune:+--tune $tune} > $R/gpurun_out/var_$name.log 2>&1
  python $R/profiles/summarize.py $(find /tmp/pv_$name -name '*.db' | head -1) | grep -E "k_dec|total kernel" > $R/gpurun_out/var_$name.txt
  echo "== $name ($tune)"; cat $R/gpurun_out/var_$name.txt | cut -c1-120
done
