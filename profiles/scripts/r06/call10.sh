#!/bin/bash
# r06 call 10: ranks sharing a device run three launches — the world-8 one-device cases repeated (the r06 flake), then the whole suite,
# smoke and the driver's bench line on the final tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
T0=$(date +%s)
for i in 1 2 3 4; do
  timeout 600 python -m pytest tests/test_comm_gpu.py tests/test_realgeom_gpu.py -m gpu -x -q -k "8 or world" > $O/call10_rep$i.log 2>&1
  echo "rep $i rc=$? $(tail -1 $O/call10_rep$i.log | cut -c1-160)" | tee -a $O/call10_status.txt
done
T1=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/call10_pytest_full.log 2>&1; echo "full rc=$? $(( $(date +%s) - T1 )) s" | tee -a $O/call10_status.txt
tail -3 $O/call10_pytest_full.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/call10_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/call10_status.txt
timeout 900 python bench.py > $O/call10_bench_driver_line.json 2> $O/call10_bench.err; echo "bench rc=$?" | tee -a $O/call10_status.txt
cut -c1-600 $O/call10_bench_driver_line.json
echo "total $(( $(date +%s) - T0 )) s"
