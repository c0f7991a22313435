#!/bin/bash
# r04, GPU call 6: 128-row block tiles in the general GEMM (encoder Linears): parity, per-shape A/B, encoder passes end to end.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run7; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm and not gemm_ps" > $O/pytest_gemm.log 2>&1; rc=$?; echo "gemm tests rc=$rc" | tee -a $O/status.txt; tail -3 $O/pytest_gemm.log
timeout 600 python profiles/bench_enc_gemm.py --iters 20 > $O/enc_gemm.jsonl 2> $O/enc_gemm.err; echo "enc gemm rc=$?" | tee -a $O/status.txt
python - <<PY
import json
for ln in open("$O/enc_gemm.jsonl"):
    d = json.loads(ln); k = list(d)[0]; r = d[k]
    print("%-10s %-18s tall0 %7.1f  tall1 %7.1f  tall2 %7.1f   ps1 %s ps2 %s  err %s" % (k, r["shape"], r["gemm_tall0"], r["gemm_tall1"], r["gemm_tall2"], r.get("ps_cfg1"), r.get("ps_cfg2"), {a: "%.1e" % b for a, b in r.items() if a.startswith("err_tall")}))
PY
for tv in 0 1 2 0 1 2; do
  timeout 400 python bench.py --layers 2 --steps 4 --warmup 2 --phase-iters 7 --no-cpu-baseline --tune gemm_tall=$tv > $O/bench_tall$tv.json 2> $O/bench_tall$tv.err
  python - <<PY
import json
d = json.loads(open("$O/bench_tall$tv.json").read().strip().splitlines()[-1])
print("gemm_tall=$tv", "vit+proj ms", d["vit_projector_ms"], "min", d["phase_min_ms"]["vit_proj_ms"], "audio ms", d["audio_encoder_ms"], "min", d["phase_min_ms"]["audio_ms"])
PY
done | tee $O/encoders_ab.txt
timeout 900 python -m pytest tests/test_realgeom_gpu.py tests/test_model_gpu.py -x -q -k "encoders or vision or audio or vit or whale" > $O/pytest_enc.log 2>&1; echo "encoder tests rc=$?" | tee -a $O/status.txt; tail -3 $O/pytest_enc.log
