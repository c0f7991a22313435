#!/bin/bash
# Build variant copies of the library for vh_gemm_ws.hip experiments into build/abl/ — run HERE (hipcc cross-compiles):
#   profiles/ablate_ws.sh name "flags" [name "flags" ...]     e.g.  fd4 "-DWS_FD=4"  noW "-DWS_ABLATE=2"
# Only the row-tile counts in $WS_RTS (default 9 and 10: uniform routing at S = 552) are instantiated.  On the GPU box:
#   VITA_AMD_LIB=build/abl/libvita_hip_<name>.so python profiles/bench_moe_gemm.py --ab 2 --nocheck
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/abl
RTS=${WS_RTS:-"WS_CASE(9) WS_CASE(10)"}
build_one() {
  n=$1; flags=$2
  mkdir -p $R/build/abl/$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags "-DWS_DEV_RTS=$RTS" -c $R/vita_amd/csrc/vh_gemm_ws.hip -o $R/build/abl/$n/ws.o -I $R/vita_amd/csrc -I $R/include -save-temps=obj 2>/dev/null
  python3 $R/profiles/audit_ws.py $R/build/abl/$n/vh_gemm_ws-hip-amdgcn-amd-amdhsa-gfx950.s | tail -1 | sed "s/^/[$n] /"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/abl/libvita_hip_$n.so $R/build/abl/$n/ws.o $R/vita_amd/lib/vh_gemm_ps.o $R/vita_amd/lib/vh_decode.o $R/vita_amd/lib/vh_gemm.o $R/vita_amd/lib/vh_attn.o $R/vita_amd/lib/vh_elem.o $R/vita_amd/lib/vh_comm.o $R/vita_amd/lib/vh_api.o -ldl
  rm -rf $R/build/abl/$n
}
while [ $# -ge 2 ]; do build_one "$1" "$2" & shift 2; done
wait
