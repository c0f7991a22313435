#!/bin/bash
# r04, GPU call 12: encoder Linear shapes on the specialised streaming GEMM with smaller m-tiles (rt_cap) and K splits.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run13; mkdir -p $O
cd $R
timeout 600 python profiles/bench_enc_sp.py --iters 20 > $O/enc_sp.jsonl 2> $O/enc_sp.err; echo "rc=$?"; tail -3 $O/enc_sp.err
grep "^##" $O/enc_sp.jsonl
