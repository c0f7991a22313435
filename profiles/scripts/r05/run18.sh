#!/bin/bash
# r05 call 18: host-side error handling of the fork / join events (vh_api.hip overlap_begin / overlap_end) — both decode schedules, the TP paths, smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_fullsize_gpu.py tests/test_comm_gpu.py tests/test_mixtral_gpu.py -m gpu -x -q -k "schedule or overlap or tp2 or tp_engine or world or decode_step or deterministic" > $O/run18_pytest.log 2>&1; echo "pytest rc=$?"
tail -2 $O/run18_pytest.log | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
