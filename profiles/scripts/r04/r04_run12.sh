#!/bin/bash
# r04, GPU call 11: flash-form attention with the next tile's staging behind the QK products: parity + timing (+ the no-staging ablations).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run12; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "attention tests rc=$?" | tee -a $O/status.txt; tail -3 $O/pytest_attn.log
timeout 900 python -m pytest tests/test_paged_gpu.py tests/test_mixtral_gpu.py -x -q -k "interleaved or alternative or group4 or real_width" > $O/pytest_engine.log 2>&1; echo "engine tests rc=$?" | tee -a $O/status.txt; tail -3 $O/pytest_engine.log
for n in 0 2 3 0; do
  lib=""; [ $n != 0 ] && lib="$R/build/abl/libvita_hip_fa_$n.so"
  echo -n "FA_ABLATE=$n "; VITA_AMD_LIB=$lib timeout 200 python profiles/bench_attn.py --only-default --rounds 1 --iters 20 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print({k: v for k, v in d.items() if k.startswith('prefill')})"
done | tee $O/fa_ablate.txt
timeout 600 python profiles/bench_attn.py --iters 30 --rounds 2 > $O/bench_attn.jsonl 2> $O/bench_attn.err; echo "bench_attn rc=$?" | tee -a $O/status.txt
grep -h "default\|attn_fa\": 0}" $O/bench_attn.jsonl
