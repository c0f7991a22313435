#!/bin/bash
# r05 call 1: the round's new parity tests on the r04 kernels + per-kernel decode times of one rank's shard at TP = 2 / 4 / 8
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
( timeout 300 python -m pytest tests/test_mixtral_gpu.py tests/test_comm_gpu.py -m gpu -x -q -k "real_width or two_ranks or rccl_binding or torch_allreduce or world_4_and_8" 2>&1 | tail -15 ) > $O/new_tests_a.txt
tail -3 $O/new_tests_a.txt
( timeout 900 python -m pytest tests/test_realgeom_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^\[stream\]" | tail -40 ) > $O/new_tests_realgeom.txt
tail -6 $O/new_tests_realgeom.txt | cut -c1-300
bash profiles/prof_tp_emulated.sh 1 2 4 8 2>&1 | tail -40
