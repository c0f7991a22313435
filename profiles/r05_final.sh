#!/bin/bash
# r05, final evidence on the final tree: the whole GPU suite (what the driver runs, < 1200 s), then the round's measurement script
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
/usr/bin/time -v timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/r05_pytest_gpu_full.log 2>&1; echo "full pytest rc=$?" | tee $O/status.txt
grep -E "passed|failed|error|Elapsed|\[realgeom\] TP" $O/r05_pytest_gpu_full.log | tail -8
bash profiles/r05_measure.sh > $O/measure.log 2>&1; echo "measure rc=$?" | tee -a $O/status.txt
head -60 $O/measure.log | cut -c1-300
