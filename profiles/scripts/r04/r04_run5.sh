#!/bin/bash
# r04, GPU call 4: parity of the ping-pong form (ps_cfg = 3), interleaved A/B 1 / 2 / 3 (uniform, skewed), schedule 0 vs 1 at lead 1,
# end-to-end bench per form.  Output: gpurun_out/r04_run5/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run5; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm_ps" > $O/pytest_gemm_ps.log 2>&1; echo "pytest gemm_ps rc=$?" | tee -a $O/status.txt
tail -2 $O/pytest_gemm_ps.log
for mode in uniform skew; do
  fl=""; [ $mode == skew ] && fl="--skew"
  timeout 400 python profiles/bench_moe_gemm.py --ab 1,2,3 --rounds 3 $fl > $O/ab123_$mode.log 2>&1; echo "ab123 $mode rc=$?" | tee -a $O/status.txt
  VITA_AMD_LIB=$R/build/abl/libvita_hip_sp_sched0.so timeout 300 python profiles/bench_moe_gemm.py --ab 2,3 --rounds 2 $fl > $O/sched0_$mode.log 2>&1; echo "sched0 $mode rc=$?" | tee -a $O/status.txt
  echo "== ab 1,2,3 $mode"; grep -h "rows per\|round\|cfg=" $O/ab123_$mode.log | cut -c1-170
  echo "== sched0 $mode"; grep -h "round" $O/sched0_$mode.log | cut -c1-170
done
for c in 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --phase-iters 5 --no-cpu-baseline --tune ps_cfg=$c > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench cfg$c rc=$?" | tee -a $O/status.txt
done
python - <<PY
import json
for c in (2, 3):
    try:
        d = json.loads(open("$O/bench_cfg%d.json" % c).read().strip().splitlines()[-1])
        print("cfg", c, "prefill_ms", d["prefill_ms"], "min", d["phase_min_ms"]["prefill_ms"], "decode", d["value"], "rf_prefill", d["roofline_prefill"]["avg_launch_us"], d["roofline_prefill"]["frac"])
    except Exception as e:
        print("cfg", c, "no line", e)
PY
