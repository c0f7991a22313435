#!/bin/bash
# r04, GPU call 18: encoder scratch kept across passes: encoder tests + the driver's bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run18; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_serving_gpu.py -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
for i in 1 2; do
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench$i.json 2> $O/bench$i.err
python - <<PY
import json
d = json.loads(open("$O/bench$i.json").read().strip().splitlines()[-1])
print("tok/s", d["value"], "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], "min", d["phase_min_ms"]["vit_proj_ms"], "aud", d["audio_encoder_ms"], "min", d["phase_min_ms"]["audio_ms"], "ttft", d["ttft_ms"])
PY
done
