#!/bin/bash
# r05 call 13: re-verify the two tests edited after the final suite run (video shape at 8 layers with the margin-aware check; TP = 2 schedules)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_video_shape_gpu.py tests/test_comm_gpu.py -m gpu -x -q -s -k "video or tp2_engine or torch_allreduce" 2>&1 | grep -v "^\[stream\]" | grep -E "router|  row |passed|failed|Error|assert|overlapped" | tail -12 ) > $O/run13_tests.txt
cat $O/run13_tests.txt | cut -c1-260
