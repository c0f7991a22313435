#!/bin/bash
# r04, GPU call 10: ablations of the flash-form attention kernel (what bounds a 64-key tile).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run11; mkdir -p $O
cd $R
for n in 0 1 2 3 4 8 16 31 0; do
  lib=""; [ $n != 0 ] && lib="$R/build/abl/libvita_hip_fa_$n.so"
  echo -n "FA_ABLATE=$n "; VITA_AMD_LIB=$lib timeout 200 python profiles/bench_attn.py --only-default --rounds 1 --iters 20 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print({k: v for k, v in d.items() if k.startswith('prefill')})"
done | tee $O/fa_ablate.txt
