#!/bin/bash
# r04, GPU call 2: the new / changed GPU tests (TP over the bulk all-reduce, TP = 8 at the released shard shapes, chunked prefill
# against the 32-layer oracle, pruned alternative paths), then the whole GPU suite if time permits.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run2; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_comm_gpu.py -x -q -s > $O/pytest_comm.log 2>&1; echo "comm rc=$?" | tee -a $O/status.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -s > $O/pytest_fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/status.txt
timeout 600 python -m pytest tests/test_mixtral_gpu.py tests/test_paged_gpu.py tests/test_ops_gpu.py -x -q > $O/pytest_engine.log 2>&1; echo "engine rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest_comm.log; grep -E "oracle:|one-shot|chunked|passed|failed" $O/pytest_fullsize.log | tail -8; tail -3 $O/pytest_engine.log
