#!/bin/bash
# r03: the register-direct weight-streaming kernel (ps_cfg=2, vh_gemm_ws.hip) against r02's register-staged one (ps_cfg=1):
# parity of every vh_gemm_ps test on all variants, then interleaved A/B rounds of the MoE pair at S = 552.
mkdir -p gpurun_out/r03
timeout 300 python3 -m pytest tests/test_ops_gpu.py -q -x -k "gemm_ps or split_planes" 2>&1 | grep -v "^  File\|Extension modules" | tail -12 > gpurun_out/r03/ws_parity.txt
cat gpurun_out/r03/ws_parity.txt
timeout 300 python3 profiles/bench_moe_gemm.py --ab 1,2 --rounds 3 --iters 10 > gpurun_out/r03/ws_ab_uniform.log 2>&1; grep -E "cfg=|rows|round|fault|Error" gpurun_out/r03/ws_ab_uniform.log
timeout 300 python3 profiles/bench_moe_gemm.py --skew --ab 1,2 --rounds 3 --iters 10 > gpurun_out/r03/ws_ab_skew.log 2>&1; grep -E "cfg=|rows|round|fault|Error" gpurun_out/r03/ws_ab_skew.log
