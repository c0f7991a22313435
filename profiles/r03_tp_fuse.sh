#!/bin/bash
# r03: cost of the tensor-parallel exchange at decode, fused into the kernels (tp_fuse=1, VhXchg) vs one all-reduce kernel
# per exchange (tp_fuse=0, r02): the whole TP=2 and TP=4 paths with all ranks sharing ONE GPU over the IPC transport
# (functional, not a scaling number: the ranks time-slice the device; the DIFFERENCE between the two settings is the
# protocol cost of 65 exchanges per token).
mkdir -p gpurun_out/r03
O=gpurun_out/r03/tp_fuse.jsonl
: > $O
for n in 2 4; do
  for f in 1 0 1 0; do
    timeout 400 python3 bench.py --gpus $n --one-device --backend gloo --collective ipc --steps 48 --warmup 8 --phase-iters 2 --no-cpu-baseline --tune tp_fuse=$f 2>gpurun_out/r03/tp_fuse.err | grep '^{' >> $O
  done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r03/tp_fuse.jsonl"):
    d = json.loads(l)
    print("tp", d["n_gpus"], d.get("tune"), d["config"]["collective"], "tok/s", d["value"], "ms/step", d["ms_per_step"], "prefill", d["prefill_ms"])
PY
tail -3 gpurun_out/r03/tp_fuse.err
