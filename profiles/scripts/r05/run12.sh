#!/bin/bash
# r05 call 12: the video-shaped prompt at 16 layers with the margin-aware router check
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_video_shape_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^\[stream\]" | tail -25 ) > $O/run12_video16.txt
grep -E "router|row |hidden|logits of|passed|failed|Error|assert" $O/run12_video16.txt | cut -c1-260
