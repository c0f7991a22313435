#!/bin/bash
# r03 A/B on one box: hi/lo splits on gfx950's v_cvt_pk_bf16_f32 (default build) against the integer-arithmetic RNE of
# r01-r02 (build/abl/libvita_hip_swcvt.so = the parent commit's sources)
for lib in "" build/abl/libvita_hip_swcvt.so "" build/abl/libvita_hip_swcvt.so; do
  VITA_AMD_LIB=$lib timeout 200 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=[$lib]', 'tok/s', d['value'], 'vit', d['vit_projector_ms'], 'aud', d['audio_encoder_ms'], 'prefill', d['prefill_ms'])"
done | tee gpurun_out/cvt_ab.txt
