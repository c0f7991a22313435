#!/bin/bash
# Build ablated copies of the library (PS_ABLATE bit mask: 1 no activation DMA, 2 no weight loads / stores, 4 no MFMAs, 8 no
# activation fragment reads, 32 no epilogue stores) with the mask applied to vh_gemm_sp.hip (default) or vh_gemm_ps.hip
# (ABL_SRC=vh_gemm_ps) into build/abl/ — run HERE (hipcc cross-compiles), then time them on the GPU box:
#   VITA_AMD_LIB=build/abl/libvita_hip_sp_<n>.so python profiles/bench_moe_gemm.py --nocheck ...
set -e
R=$(cd $(dirname $0)/.. && pwd)
SRC=${ABL_SRC:-vh_gemm_sp}
mkdir -p $R/build/abl
for n in "$@"; do
  (
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPS_ABLATE=$n ${ABL_DEFS} -c $R/vita_amd/csrc/$SRC.hip -o $R/build/abl/${SRC}_$n.o -I $R/vita_amd/csrc -I $R/include -Wno-unused-result
  objs=""
  for o in vh_decode vh_gemm vh_gemm_ps vh_gemm_sp vh_attn vh_elem vh_comm vh_api; do
    if [ $o == $SRC ]; then objs="$objs $R/build/abl/${SRC}_$n.o"; else objs="$objs $R/vita_amd/lib/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/abl/libvita_hip_${SRC#vh_gemm_}_$n.so $objs -ldl
  ) &
done
wait
ls -la $R/build/abl/*.so
