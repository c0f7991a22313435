#!/bin/bash
# r04, final evidence on the final tree: the round's measurement script, the whole GPU suite, the reference's assets at 32 layers.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_final; mkdir -p $O $R/gpurun_out/r04
cd $R
bash profiles/r04_measure.sh > $O/measure.log 2>&1; echo "measure rc=$?" | tee -a $O/status.txt
head -12 $O/measure.log | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r04/r04_pytest_gpu_full.log 2>&1; echo "full pytest rc=$?" | tee -a $O/status.txt
tail -4 $R/gpurun_out/r04/r04_pytest_gpu_full.log
VITA_ASSETS_LAYERS=32 timeout 900 python -m pytest tests/test_assets_gpu.py -x -q -s > $R/gpurun_out/r04/r04_assets_parity_32_layers.txt 2>&1; echo "assets 32 layers rc=$?" | tee -a $O/status.txt
grep -h "\[assets\]\|passed\|failed\|max\|err" $R/gpurun_out/r04/r04_assets_parity_32_layers.txt | cut -c1-200 | tail -20
