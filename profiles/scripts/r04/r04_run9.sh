#!/bin/bash
# r04, GPU call 8: flash-form attention with two key halves per block (8 waves): parity, micro-benchmark, phases end to end.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run9; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "attention tests rc=$?" | tee -a $O/status.txt; tail -5 $O/pytest_attn.log
timeout 900 python -m pytest tests/test_paged_gpu.py tests/test_mixtral_gpu.py -x -q -k "interleaved or alternative or group4 or real_width" > $O/pytest_engine.log 2>&1; echo "engine tests rc=$?" | tee -a $O/status.txt; tail -5 $O/pytest_engine.log
timeout 600 python profiles/bench_attn.py --iters 30 --rounds 2 > $O/bench_attn.jsonl 2> $O/bench_attn.err; echo "bench_attn rc=$?" | tee -a $O/status.txt
cat $O/bench_attn.jsonl
for fa in 0 1 0 1; do
  timeout 400 python bench.py --layers 8 --steps 4 --warmup 2 --phase-iters 7 --no-cpu-baseline --tune attn_fa=$fa > $O/bench_fa$fa.json 2> $O/bench_fa$fa.err
  python - <<PY
import json
d = json.loads(open("$O/bench_fa$fa.json").read().strip().splitlines()[-1])
print("attn_fa=$fa", "prefill(8 layers) ms", d["prefill_ms"], "min", d["phase_min_ms"]["prefill_ms"], "vit+proj ms", d["vit_projector_ms"], "min", d["phase_min_ms"]["vit_proj_ms"], "audio", d["audio_encoder_ms"], "tok/s", d["value"])
PY
done | tee $O/phases_ab.txt
