#!/bin/bash
# dispatch-by-dispatch trace of one ViT+projector pass and one Whale pass: prof_encoders.sh name[:tune]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%%:*}; tune=""; [[ "$spec" == *:* ]] && tune=${spec#*:}
  rm -rf /tmp/pe_$name
  timeout ${PP_TIMEOUT:-120} rocprofv3 --kernel-trace -d /tmp/pe_$name -o r -- python $R/bench.py --layers 1 --steps 2 --warmup 1 --phase-warmup 1 --phase-iters 2 --no-cpu-baseline ${tune:+--tune $tune} > $R/gpurun_out/pe_$name.log 2>&1
  db=$(find /tmp/pe_$name -name '*.db' | head -1)
  python $R/profiles/layer_trace.py $db k_vit_patchify 2 k_embed_splice > $R/gpurun_out/enc_$name.txt
  echo "== $name ($tune)"; tail -1 $R/gpurun_out/pe_$name.log | cut -c1-100; wc -l $R/gpurun_out/enc_$name.txt
done
