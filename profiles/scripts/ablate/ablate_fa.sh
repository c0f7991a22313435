#!/bin/bash
# Build ablated copies of the library for the flash-form attention kernel (FA_ABLATE bit mask, vh_attn.hip) into build/abl/ — run HERE
# (hipcc cross-compiles), then time them on the GPU box:  VITA_AMD_LIB=build/abl/libvita_hip_fa_<n>.so python profiles/bench_attn.py --only-default
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/abl
for n in "$@"; do
  (
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFA_ABLATE=$n -c $R/vita_amd/csrc/vh_attn.hip -o $R/build/abl/vh_attn_$n.o -I $R/vita_amd/csrc -I $R/include -Wno-unused-result
  objs=""
  for o in vh_decode vh_gemm vh_gemm_ps vh_gemm_sp vh_attn vh_elem vh_comm vh_api; do
    if [ $o == vh_attn ]; then objs="$objs $R/build/abl/vh_attn_$n.o"; else objs="$objs $R/vita_amd/lib/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/abl/libvita_hip_fa_$n.so $objs -ldl
  ) &
done
wait
ls $R/build/abl/*.so
