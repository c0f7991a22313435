#!/bin/bash
# r04, GPU call 22: split-K reducer + LayerNorm with one wave per row (N <= 1024): encoder parity + A/B of the encoder phases
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run22; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -k "vit or vision or tower or gemm or encoder or whale or audio or layernorm" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
for fr in 1 8; do
for rw in 0 1 0 1; do
  timeout 400 python bench.py --layers 2 --steps 4 --warmup 2 --phase-iters 7 --no-cpu-baseline --frames $fr --tune reduce_wave=$rw > $O/bench_f${fr}_rw$rw.json 2> $O/bench_f${fr}_rw$rw.err
  python - <<PY
import json
d = json.loads(open("$O/bench_f${fr}_rw$rw.json").read().strip().splitlines()[-1])
print("frames=$fr reduce_wave=$rw", "vit+proj ms", d["vit_projector_ms"], "min", d["phase_min_ms"]["vit_proj_ms"], "audio", d["audio_encoder_ms"], "min", d["phase_min_ms"]["audio_ms"])
PY
done; done | tee $O/reduce_wave_ab.txt
