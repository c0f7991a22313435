#!/bin/bash
# r04, GPU call 3: stager lead 2 (default) vs 1, cost-ordered grid-stride schedule vs per-XCD runs (uniform and skewed routing),
# end-to-end bench with the new default, the 8-frame video-shaped parity test and bench line.  Output: gpurun_out/r04_run4/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run4; mkdir -p $O
cd $R
for mode in uniform skew; do
  fl=""; [ $mode == skew ] && fl="--skew"
  timeout 300 python profiles/bench_moe_gemm.py --ab 1,2 --rounds 2 $fl > $O/default_$mode.log 2>&1; echo "default $mode rc=$?" | tee -a $O/status.txt
  for v in lead1 sched1; do
    VITA_AMD_LIB=$R/build/abl/libvita_hip_sp_$v.so timeout 300 python profiles/bench_moe_gemm.py --ab 2 --rounds 2 $fl > $O/${v}_$mode.log 2>&1; echo "$v $mode rc=$?" | tee -a $O/status.txt
  done
done
for mode in uniform skew; do for v in default lead1 sched1; do echo "== $v $mode"; grep -h "rows per\|round" $O/${v}_$mode.log | cut -c1-170; done; done
timeout 600 python bench.py --steps 20 --warmup 5 --phase-iters 5 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/status.txt
VITA_AMD_LIB=$R/build/abl/libvita_hip_sp_sched1.so timeout 600 python bench.py --steps 20 --warmup 5 --phase-iters 5 --no-cpu-baseline > $O/bench_sched1.json 2> $O/bench_sched1.err; echo "bench sched1 rc=$?" | tee -a $O/status.txt
python - <<PY
import json
for c in ("default", "sched1"):
    try:
        d = json.loads(open("$O/bench_%s.json" % c).read().strip().splitlines()[-1])
        print(c, "prefill_ms", d["prefill_ms"], "min", d["phase_min_ms"]["prefill_ms"], "decode", d["value"], "rf_prefill", d["roofline_prefill"]["avg_launch_us"], d["roofline_prefill"]["frac"], "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"])
    except Exception as e:
        print(c, "no line", e)
PY
timeout 900 python bench.py --frames 8 --steps 20 --warmup 5 --phase-iters 3 --no-cpu-baseline > $O/bench_frames8.json 2> $O/bench_frames8.err; echo "bench frames8 rc=$?" | tee -a $O/status.txt
timeout 1500 python -m pytest tests/test_video_shape_gpu.py -x -q -s > $O/pytest_video.log 2>&1; echo "video test rc=$?" | tee -a $O/status.txt
grep -E "\[video\]|router|logits|passed|failed|Error" $O/pytest_video.log | tail -12
