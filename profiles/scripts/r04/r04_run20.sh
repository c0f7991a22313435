#!/bin/bash
# r04, GPU call 20: the whole GPU suite on the final tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r04
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r04/r04_pytest_gpu_full.log 2>&1; echo "full pytest rc=$?"
tail -4 $R/gpurun_out/r04/r04_pytest_gpu_full.log
