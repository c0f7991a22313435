#!/bin/bash
# r06 call 15: the whole GPU suite once more on the committed tree (flake check of the multi-process cases), then the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/call15_pytest_full.log 2>&1; echo "full rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call15_status.txt
tail -3 $O/call15_pytest_full.log | cut -c1-300
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/call15_bench.json 2> $O/call15_bench.err; echo "bench rc=$?" | tee -a $O/call15_status.txt
cut -c1-400 $O/call15_bench.json
echo "total $(( $(date +%s) - T0 )) s"
