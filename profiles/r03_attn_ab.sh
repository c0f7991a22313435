#!/bin/bash
# r03 A/B on one box: plain / causal attention on bf16 x 3 MFMAs (default) against the fp32-MFMA direct kernel (attn_impl=2)
for t in "" "attn_impl=2" "" "attn_impl=2"; do
  timeout 200 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 ${t:+--tune $t} 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tune=[$t]', 'vit', d['vit_projector_ms'], 'aud', d['audio_encoder_ms'], 'prefill', d['prefill_ms'], 'min', d['phase_min_ms'])"
done | tee gpurun_out/attn_ab.txt
