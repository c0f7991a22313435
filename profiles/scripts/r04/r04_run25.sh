#!/bin/bash
# r04, GPU call 25: the specialised GEMM with plain weight loads as the default: streaming-GEMM tests, the 32-layer parity test, the driver's line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run25; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_realgeom_gpu.py tests/test_mixtral_gpu.py -x -q -k "gemm_ps or backbone or real_width or group4 or tiny" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("tok/s", d["value"], "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"], "ttft", d["ttft_ms"], "rf_prefill", d["roofline_prefill"]["avg_launch_us"], d["roofline_prefill"]["frac"], d["roofline_prefill"]["traffic"])
PY
