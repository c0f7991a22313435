#!/bin/bash
# dispatch-by-dispatch trace of one prefill layer: prof_layer.sh name[:tune] ...   (-> gpurun_out/layer_<name>.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%%:*}; tune=""; [[ "$spec" == *:* ]] && tune=${spec#*:}
  rm -rf /tmp/pl_$name
  timeout ${PP_TIMEOUT:-120} rocprofv3 --kernel-trace -d /tmp/pl_$name -o r -- python $R/bench.py --layers ${PP_LAYERS:-4} --steps 4 --warmup 2 --phase-iters 2 --no-cpu-baseline ${tune:+--tune $tune} > $R/gpurun_out/pl_$name.log 2>&1
  db=$(find /tmp/pl_$name -name '*.db' | head -1)
  python $R/profiles/layer_trace.py $db > $R/gpurun_out/layer_$name.txt
  python $R/profiles/summarize.py $db > $R/gpurun_out/stats_$name.txt
  echo "== $name ($tune)"; cat $R/gpurun_out/layer_$name.txt | cut -c1-150
done
