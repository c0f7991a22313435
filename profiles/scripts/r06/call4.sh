#!/bin/bash
# r06 call 4: comm tests with per-rank diagnostics; k_dec_ablk at 4 blocks per CU (V tile shares the K buffer); XCD-contiguous tile
# placement of the streaming GEMM (microbench A/B, uniform and skewed routing)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests/test_comm_gpu.py -m gpu -q --durations=12 > $O/call4_pytest_comm.log 2>&1; echo "pytest comm rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call4_status.txt
tail -12 $O/call4_pytest_comm.log | cut -c1-300
timeout 600 python -m pytest tests/test_mixtral_gpu.py tests/test_fullsize_gpu.py tests/test_ops_gpu.py -m gpu -x -q > $O/call4_pytest_b.log 2>&1; echo "pytest b rc=$? $(( $(date +%s) - T0 )) s" | tee -a $O/call4_status.txt
tail -3 $O/call4_pytest_b.log | cut -c1-300
timeout 600 python3 profiles/bench_moe_gemm.py --abtune ps_xcd=0,1,3 --rounds 3 > $O/c4_moe_xcd_uniform.txt 2>&1; tail -12 $O/c4_moe_xcd_uniform.txt | cut -c1-200
timeout 600 python3 profiles/bench_moe_gemm.py --skew --abtune ps_xcd=0,1,3 --rounds 3 > $O/c4_moe_xcd_skew.txt 2>&1; tail -12 $O/c4_moe_xcd_skew.txt | cut -c1-200
B="python3 bench.py --no-cpu-baseline --phase-iters 3 --phase-warmup 1"
for rep in 1 2; do
  timeout 300 $B --steps 40 --warmup 5 > $O/c4_tp1_r$rep.json 2>> $O/c4.err
  timeout 300 $B --steps 40 --warmup 5 --tune dec_fused=0 > $O/c4_tp1_3launch_r$rep.json 2>> $O/c4.err
done
for tp in 8 4 2; do
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp > $O/c4_emu${tp}_skip.json 2>> $O/c4.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --tune dec_fused=0 > $O/c4_emu${tp}_skip_3launch.json 2>> $O/c4.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange fused > $O/c4_emu${tp}_loop_fused.json 2>> $O/c4.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange kernel > $O/c4_emu${tp}_loop_kernel.json 2>> $O/c4.err
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c4_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "ms", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], d["config"].get("decode_schedule"), d["config"].get("collective"),
              "gateup us", d["roofline"]["avg_launch_us"], "prefill", d["prefill_ms"], d["phase_min_ms"]["prefill_ms"], (d.get("emulated_tp") or {}).get("comm_status"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
tail -5 $O/c4.err | cut -c1-300
echo "total $(( $(date +%s) - T0 )) s"
