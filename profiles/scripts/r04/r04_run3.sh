#!/bin/bash
# r04, GPU call 2: specialised streaming GEMM on the projections and end to end, static priority of its MFMA waves, LDS-read
# ablations; then the new / changed GPU tests.  Output under gpurun_out/r04_run3/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run3; mkdir -p $O
cd $R
timeout 300 python profiles/bench_proj_gemm.py --ab 1,2 --check > $O/proj_ab.json 2> $O/proj_ab.log; echo "proj rc=$?" | tee -a $O/status.txt
for v in prio1 prio3; do
  VITA_AMD_LIB=$R/build/abl/libvita_hip_sp_$v.so timeout 300 python profiles/bench_moe_gemm.py --ab 1,2 --rounds 2 > $O/sp_$v.log 2>&1; echo "$v rc=$?" | tee -a $O/status.txt
done
for n in 8 15 11; do
  VITA_AMD_LIB=$R/build/abl/libvita_hip_sp_$n.so timeout 300 python profiles/bench_moe_gemm.py --nocheck --ab 2 --rounds 2 > $O/abl_sp_$n.log 2>&1; echo "ablate $n rc=$?" | tee -a $O/status.txt
done
timeout 600 python bench.py --steps 20 --warmup 5 --phase-iters 5 --no-cpu-baseline > $O/bench_cfg1.json 2> $O/bench_cfg1.err; echo "bench cfg1 rc=$?" | tee -a $O/status.txt
timeout 600 python bench.py --steps 20 --warmup 5 --phase-iters 5 --no-cpu-baseline --tune ps_cfg=2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?" | tee -a $O/status.txt
python - <<PY
import json
for c in (1, 2):
    try:
        d = json.loads(open("$O/bench_cfg%d.json" % c).read().strip().splitlines()[-1])
        print("cfg", c, "prefill_ms", d["prefill_ms"], "min", d["phase_min_ms"]["prefill_ms"], "decode", d["value"], "rf_prefill", d["roofline_prefill"]["avg_launch_us"], d["roofline_prefill"]["frac"])
    except Exception as e:
        print("cfg", c, "no line", e)
PY
cat $O/proj_ab.log | tail -12
for v in prio1 prio3; do echo "== $v"; grep -h "round" $O/sp_$v.log | cut -c1-170; done
for n in 8 15 11; do echo "ABLATE $n"; grep -h "round" $O/abl_sp_$n.log | cut -c1-170; done
timeout 900 python -m pytest tests/test_comm_gpu.py -x -q -s > $O/pytest_comm.log 2>&1; echo "comm rc=$?" | tee -a $O/status.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -s > $O/pytest_fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/status.txt
timeout 600 python -m pytest tests/test_mixtral_gpu.py tests/test_paged_gpu.py tests/test_ops_gpu.py -x -q > $O/pytest_engine.log 2>&1; echo "engine rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest_comm.log; grep -E "oracle:|one-shot|chunked|passed|failed" $O/pytest_fullsize.log | tail -8; tail -3 $O/pytest_engine.log
