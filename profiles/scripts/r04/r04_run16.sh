#!/bin/bash
# r04, GPU call 16: smoke() and the default bench line (cpu_baseline with the encoder thread sweep) on the final tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run16; mkdir -p $O
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $O/smoke.log
timeout 900 python bench.py > $O/r04_bench_tp1.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/r04_bench_tp1.json").read().strip().splitlines()[-1])
print("tok/s", d["value"], "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"], "ttft", d["ttft_ms"])
c = d["cpu_baseline"]; print({k: c[k] for k in c if k in ("value", "cores", "prefill_ms", "vit_projector_ms", "audio_encoder_ms", "encoder_cores", "audio_encoder_cores", "encoder_ms_by_threads")})
print(d["roofline_prefill"]["kernel"][:60], d["roofline_prefill"]["avg_launch_us"], d["roofline_prefill"]["frac"])
PY
