#!/bin/bash
# r03: what do the epilogue stores of the K-split projections cost, and is that where "slow" boxes lose their time?
# old = r02 epilogue (lane-owned 16 B stores, 16 rows per wave instruction), 0 = row-contiguous stores through LDS,
# 32 = no stores at all (PS_ABLATE bit 32).  Built by profiles/ablate_ps.sh + the HEAD~ source for `old`.
mkdir -p gpurun_out/r03
for n in old 0 32 old 0 32; do
  echo -n "[epilogue=$n] "; VITA_AMD_LIB=build/abl/libvita_hip_$n.so timeout 120 python3 profiles/bench_proj_gemm.py 2>&1 | grep "^{"
done | tee gpurun_out/r03/proj_probe.txt
for n in old 0 old 0; do
  echo "[epilogue=$n]"; VITA_AMD_LIB=build/abl/libvita_hip_$n.so timeout 200 python3 profiles/bench_moe_gemm.py --skew --iters 8 2>&1 | grep -E "^stream"
  VITA_AMD_LIB=build/abl/libvita_hip_$n.so timeout 200 python3 profiles/bench_moe_gemm.py --iters 8 2>&1 | grep -E "^stream"
done | tee -a gpurun_out/r03/proj_probe.txt
