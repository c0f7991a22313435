#!/bin/bash
# r06 call 8: the whole GPU suite on the final tree (what the driver runs; budget 1200 s) + smoke()
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $O/r06_pytest_gpu_full.log 2>&1; echo "full pytest rc=$? in $(( $(date +%s) - T0 )) s" | tee $O/call8_status.txt
grep -E "passed|failed|error" $O/r06_pytest_gpu_full.log | tail -5 | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r06_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/call8_status.txt; tail -3 $O/r06_smoke.log | cut -c1-300
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/c8_driver_line.json 2> $O/c8.err
python3 -c "
import json; d=json.loads(open('$O/c8_driver_line.json').read().strip().splitlines()[-1]); print(d['value'], d['prefill_ms'], d['encode_ms'], d['ttft_ms'], d['roofline']['frac'], d['roofline']['frac_kernel_trace'], d['roofline_prefill']['frac'], d['roofline_prefill']['frac_kernel_trace'], d['roofline']['traffic_source'])"
echo "total $(( $(date +%s) - T0 )) s"
