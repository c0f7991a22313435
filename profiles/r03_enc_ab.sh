#!/bin/bash
# r03 A/B on one box: encoders with one library call per transformer block (default) against one call per operator
# (VITA_AMD_PER_OPERATOR=1), bench.py phase timings
for t in 0 1 0 1; do
  VITA_AMD_PER_OPERATOR=$t timeout 200 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('per_operator=$t', 'vit', d['vit_projector_ms'], 'aud', d['audio_encoder_ms'], 'prefill', d['prefill_ms'], 'min', d['phase_min_ms'])"
done | tee gpurun_out/enc_ab.txt
