#!/bin/bash
# r03 A/B inside one gpurun call (same box): slab sums / combine / plane split folded into the norm and attention kernels
# (default) against the separate launches (prefill_fuse_rows=0).  Prints prefill ms (median, min) per run.
mkdir -p gpurun_out/r03
for t in "" "prefill_fuse_rows=0" "" "prefill_fuse_rows=0"; do
  timeout 200 python3 bench.py --no-cpu-baseline --steps 16 --warmup 4 ${t:+--tune $t} 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tune=[$t]', 'tok/s', d['value'], 'prefill_ms', d['prefill_ms'], 'min', d['phase_min_ms']['prefill_ms'], 'vit', d['vit_projector_ms'], 'aud', d['audio_encoder_ms'])"
done | tee gpurun_out/r03/fuse_ab.txt
