#!/bin/bash
# r04, GPU call 23: the library rebuilt from the last commit: smoke() + a quick slice of the suite + the driver's bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run23; mkdir -p $O
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_mixtral_gpu.py tests/test_model_gpu.py -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("tok/s", d["value"], "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"], "ttft", d["ttft_ms"], "roofline", d["roofline"]["frac"], "rf_prefill", d["roofline_prefill"]["frac"])
PY
