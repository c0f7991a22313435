#!/bin/bash
# r03: is prefill time a function of the workspace placement or of the box?  One box, one build: the driver's exact line,
# the default, a short run, a sweep of a deliberate 64-KB-unit pad in front of the workspace; SMI clocks / power sampled
# while each runs.  Results: profiles/r03_layout_box_a.jsonl (52.5 ms under every setting), r03_layout_box_b.jsonl (40.0 ms
# under both step settings, same build): the r02 "39.7 vs 52.6 ms" was box-to-box, not placement.
mkdir -p gpurun_out/r03
O=gpurun_out/r03/layout.jsonl
: > $O
smi() { rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor junction" | tr '\n' ';' ; echo; }
run() { ( sleep 14; echo "during [$*]: $(smi)" >> gpurun_out/r03/smi.txt ) & python3 bench.py --gpus 1 --no-cpu-baseline "$@" 2>gpurun_out/r03/layout.err | grep '^{' >> $O; wait; }
echo "idle: $(smi)" > gpurun_out/r03/smi.txt
run --steps 20 --warmup 5
run --steps 64 --warmup 8
run --steps 8 --warmup 2
for pad in 1 4 5 16 33; do run --steps 20 --warmup 5 --tune ws_pad=$pad; done
python3 - <<'PY'
import json
for l in open("gpurun_out/r03/layout.jsonl"):
    d = json.loads(l)
    print(d["steps"], d["warmup"], d.get("tune"), "tok/s", d["value"], "prefill", d["prefill_ms"], d["phase_min_ms"]["prefill_ms"],
          "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"])
PY
cat gpurun_out/r03/smi.txt
