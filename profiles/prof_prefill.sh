#!/bin/bash
# per-kernel prefill timing for tuning knobs: prof_prefill.sh name:tune ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%%:*}; tune=${spec#*:}
  rm -rf /tmp/pp_$name
  timeout ${PP_TIMEOUT:-90} rocprofv3 --kernel-trace -d /tmp/pp_$name -o r -- python $R/bench.py --layers ${PP_LAYERS:-2} --steps 4 --warmup 2 --no-cpu-baseline ${tune:+--tune $tune} > $R/gpurun_out/pp_$name.log 2>&1
  echo "== $name ($tune)"
  python $R/profiles/summarize.py $(find /tmp/pp_$name -name '*.db' | head -1) | grep -E "k_gemm|k_attn<128|k_split" | cut -c1-130 | tee $R/gpurun_out/pp_$name.txt
done
