#!/bin/bash
# r05 call 10: the overlapped decode schedule under tensor parallelism (granule-in / granule-out attention all-reduce), world 2 on one device
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_comm_gpu.py tests/test_mixtral_gpu.py -m gpu -x -q -k "tp2 or world_4_and_8 or ipc_allreduce or torch_allreduce or released or two_ranks" 2>&1 | tail -8 ) > $O/run10_tests.txt
tail -4 $O/run10_tests.txt | cut -c1-300
timeout 600 python bench.py --gpus 2 --one-device --backend gloo --steps 16 --warmup 4 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline > $O/run10_bench_tp2_one_device.json 2> $O/run10_tp2.err
tail -1 $O/run10_bench_tp2_one_device.json | cut -c1-700
timeout 600 python bench.py --gpus 2 --one-device --backend gloo --steps 16 --warmup 4 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --tune dec_overlap=0 > $O/run10_bench_tp2_one_device_one_stream.json 2> $O/run10_tp2b.err
tail -1 $O/run10_bench_tp2_one_device_one_stream.json | cut -c1-400
