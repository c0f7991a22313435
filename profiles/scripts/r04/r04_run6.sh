#!/bin/bash
# r04, GPU call 5: fence-free split-KV hand-off of the decode attention: decode tests, A/B of the bench against the previous
# vh_decode build; then (only if the tests pass) the round's evidence (profiles/r04_measure.sh) and the whole GPU suite.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run6; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_mixtral_gpu.py tests/test_paged_gpu.py tests/test_edge_gpu.py tests/test_ops_gpu.py -x -q > $O/pytest_decode.log 2>&1; rc=$?; echo "decode tests rc=$rc" | tee -a $O/status.txt; tail -3 $O/pytest_decode.log
if [ $rc -ne 0 ]; then echo "decode tests failed: stopping"; exit 1; fi
for v in new old new old; do
  lib=""; [ $v == old ] && lib=$R/build/abl/libvita_hip_olddec.so
  VITA_AMD_LIB=$lib timeout 400 python bench.py --steps 64 --warmup 8 --phase-iters 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("$v", "tok/s", d["value"], "ms/step", d["ms_per_step"], "gateup us", d["roofline"]["avg_launch_us"], "prefill", d["prefill_ms"])
PY
done | tee $O/decode_ab.txt
bash profiles/r04_measure.sh > $O/measure.log 2>&1; echo "measure rc=$?" | tee -a $O/status.txt
tail -120 $O/measure.log | cut -c1-170
timeout 1500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r04/r04_pytest_gpu_full.log 2>&1; echo "full pytest rc=$?" | tee -a $O/status.txt
tail -4 $R/gpurun_out/r04/r04_pytest_gpu_full.log
