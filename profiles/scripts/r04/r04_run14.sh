#!/bin/bash
# r04, GPU call 13: ViT blocks in planes mode (streaming GEMM on bf16 hi/lo planes, one-round tilings): parity + A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run14; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -k "vit or vision or tower or gemm or attention or encoder or whale or audio" > $O/pytest_a.log 2>&1; echo "model/ops tests rc=$?" | tee -a $O/status.txt; tail -3 $O/pytest_a.log
timeout 900 python -m pytest tests/test_realgeom_gpu.py tests/test_assets_gpu.py -x -q -k "encoders" > $O/pytest_b.log 2>&1; echo "realgeom/assets encoder tests rc=$?" | tee -a $O/status.txt; tail -3 $O/pytest_b.log
for fr in 1 8; do
for pl in 0 1 0 1; do
  VITA_AMD_VIT_PLANES=$pl timeout 400 python bench.py --layers 2 --steps 4 --warmup 2 --phase-iters 7 --no-cpu-baseline --frames $fr > $O/bench_f${fr}_pl$pl.json 2> $O/bench_f${fr}_pl$pl.err
  python - <<PY
import json
d = json.loads(open("$O/bench_f${fr}_pl$pl.json").read().strip().splitlines()[-1])
print("frames=$fr planes=$pl", "vit+proj ms", d["vit_projector_ms"], "min", d["phase_min_ms"]["vit_proj_ms"], "audio", d["audio_encoder_ms"])
PY
done; done | tee $O/vit_planes_ab.txt
