#!/bin/bash
# r03: timing of the vh_gemm_ws.hip variants built by profiles/ablate_ws.sh (uniform routing, rt 9-10), one process each,
# two rounds of cfg=2 (+ one of cfg=1 in the first process as the in-box reference)
mkdir -p gpurun_out/r03
O=gpurun_out/r03/ws_ablate.txt
: > $O
first=1
for n in "$@"; do
  ab=2; [ $first = 1 ] && ab=1,2; first=0
  nc=""; case $n in *no*|*only*|hiA*) nc="--nocheck";; esac       # ablated builds compute wrong results by construction
  VITA_AMD_LIB=build/abl/libvita_hip_$n.so timeout 200 python3 profiles/bench_moe_gemm.py --ab $ab --rounds ${ROUNDS:-2} --iters 8 $nc $EXTRA 2>&1 | grep -E "^round|^cfg=|fault|parity|Error" | sed "s/^/[$n] /" >> $O
done
cat $O
