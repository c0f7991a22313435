#!/bin/bash
# r05 call 15: where does an iteration of 3 concurrent sequences spend its time (kernel table), and do the streaming GEMM's small-M knobs move it
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python3 bench.py --steps 32 --warmup 4 --no-cpu-baseline --phase-iters 1 --phase-warmup 1 --batch 3 > $O/kt_b3.json 2> $O/kt_b3.err)
db=$(find /tmp/kt -name '*.db' | head -1)
python3 $R/profiles/summarize.py $db 'anonymous namespace' 'k_fill_hash|k_gemm_sp|k_attn_fa|k_attn_x3' > $O/r05_kernel_stats_concurrent_b3.txt
head -30 $O/r05_kernel_stats_concurrent_b3.txt | cut -c1-170
python3 $R/profiles/layer_trace.py $db "k_gather_rows" 40 | cut -c1-150 > $O/r05_iteration_layer_trace_b3.txt
head -40 $O/r05_iteration_layer_trace_b3.txt
for kv in ps_nt=0 ps_nt=1 moe_ksplit=1 moe_ksplit=2 moe_ksplit=4; do
  (cd $R && timeout 200 python3 bench.py --steps 32 --warmup 4 --no-cpu-baseline --phase-iters 1 --phase-warmup 1 --batch 3,4,8 --tune $kv > $O/run15_$kv.json 2> $O/run15_$kv.err)
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/run15_$kv.json").read().strip().splitlines()[-1])
    print("$kv", [(c["batch"], c["ms_per_iteration"]) for c in d.get("concurrent", [])])
except Exception as e:
    print("$kv", "no line:", e)
PY
done
