#!/bin/bash
# Build ablated copies of the library (PS_ABLATE bit mask, see vh_gemm_ps.hip) into build/abl/ — run HERE (hipcc
# cross-compiles), then time them on the GPU box: for n in ...; do VITA_AMD_LIB=build/abl/libvita_hip_$n.so python profiles/bench_moe_gemm.py; done
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPS_ABLATE=$n -c $R/vita_amd/csrc/vh_gemm_ps.hip -o $R/build/abl/ps_$n.o -I $R/vita_amd/csrc -I $R/include
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/abl/libvita_hip_$n.so $R/build/abl/ps_$n.o $R/vita_amd/lib/vh_gemm_ws.o $R/vita_amd/lib/vh_decode.o $R/vita_amd/lib/vh_gemm.o $R/vita_amd/lib/vh_attn.o $R/vita_amd/lib/vh_elem.o $R/vita_amd/lib/vh_comm.o $R/vita_amd/lib/vh_api.o -ldl
done
