#!/bin/bash
# r04, GPU call 17: the timed decode-exchange trial (what a real node runs at bring-up) forced on one device + the TP suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run17; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_comm_gpu.py -x -q -s -k "tp2_engine" > $O/pytest_tp2.log 2>&1; echo "tp2 tests rc=$?"; grep -h "decode exchange\|passed\|failed\|Error\|error" $O/pytest_tp2.log | tail -12
