#!/bin/bash
# r05 call 17: final tree after the small-M fixes (the specialised kernel's code is byte-identical to the tree of r05_pytest_gpu_full.log; the 8-wave
# kernels are those of call 16's 197-test pass): streaming-GEMM / paged / full-size tests again, then the concurrency line that replaces r05_bench_concurrent.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_ops_gpu.py tests/test_paged_gpu.py tests/test_fullsize_gpu.py tests/test_serving_gpu.py -m gpu -x -q -k "gemm_ps or paged or fullsize or serving" > $O/run17_pytest.log 2>&1; echo "pytest rc=$?"
tail -2 $O/run17_pytest.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 400 python3 bench.py --steps 24 --warmup 4 --no-cpu-baseline --batch 2,3,4,8,16 > $O/r05_bench_concurrent.json 2> $O/run17_bench.err)
python3 - <<PY
import json
try:
    d = json.loads(open("$O/r05_bench_concurrent.json").read().strip().splitlines()[-1])
    print("tok/s", d["value"], "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"], "roofline", d["roofline"]["frac"], "rf_prefill", d["roofline_prefill"]["avg_launch_us"])
    for c in d.get("concurrent", []): print("   B", c["batch"], c["aggregate_tokens_per_s"], c["ms_per_iteration"])
except Exception as e:
    print("no line:", e)
PY
