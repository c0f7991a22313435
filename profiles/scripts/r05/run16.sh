#!/bin/bash
# r05 call 16: the small-M fixes of the streaming GEMM (K-split estimator without the M-split credit / with the slab price scaled by the rows when every
# expert holds one row tile; non-temporal weight loads there): the GPU suite without its three longest tests, then the concurrency line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 700 python -m pytest tests -m gpu -x -q --durations=8 \
  --deselect tests/test_realgeom_gpu.py::test_tp8_full_depth_omni_matches_oracle \
  --deselect tests/test_video_shape_gpu.py::test_eight_frame_video_prompt_matches_oracle \
  --deselect tests/test_assets_gpu.py > $O/run16_pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s"
grep -E "passed|failed|error" $O/run16_pytest.log | tail -5 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 400 python3 bench.py --steps 24 --warmup 4 --no-cpu-baseline --batch 2,3,4,8,16 > $O/run16_bench_concurrent.json 2> $O/run16_bench.err)
python3 - <<PY
import json
try:
    d = json.loads(open("$O/run16_bench_concurrent.json").read().strip().splitlines()[-1])
    print("tok/s", d["value"], "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"], "roofline", d["roofline"]["frac"], "rf_prefill", d["roofline_prefill"]["avg_launch_us"])
    for c in d.get("concurrent", []): print("   B", c["batch"], c["aggregate_tokens_per_s"], c["ms_per_iteration"])
except Exception as e:
    print("no line:", e)
PY
