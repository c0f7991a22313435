#!/bin/bash
# r06 call 1: the fused attention-block launch (k_dec_ablk) — targeted parity tests, then A/B of the decode schedules at TP = 1 and on
# one rank's TP = 2 / 4 / 8 shard with the exchanges skipped / looped back (fused and kernel forms)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 900 python -m pytest tests/test_mixtral_gpu.py tests/test_fullsize_gpu.py tests/test_ops_gpu.py tests/test_edge_gpu.py -m gpu -x -q --durations=8 > $O/call1_pytest_a.log 2>&1; echo "pytest a rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call1_status.txt
tail -4 $O/call1_pytest_a.log | cut -c1-300
timeout 900 python -m pytest tests/test_comm_gpu.py tests/test_paged_gpu.py -m gpu -x -q --durations=8 > $O/call1_pytest_b.log 2>&1; echo "pytest b rc=$? $(( $(date +%s) - T0 )) s" | tee -a $O/call1_status.txt
tail -4 $O/call1_pytest_b.log | cut -c1-300
B="python3 bench.py --no-cpu-baseline --phase-iters 1 --phase-warmup 1"
for f in -1 0; do
  timeout 300 $B --steps 40 --warmup 5 --tune dec_fused=$f > $O/tp1_fused$f.json 2> $O/tp1_fused$f.err
done
for tp in 8 4 2; do
  for f in -1 0; do
    timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --tune dec_fused=$f > $O/emu${tp}_skip_fused$f.json 2> $O/emu.err
  done
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange fused > $O/emu${tp}_loop_fused.json 2> $O/emu.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange kernel > $O/emu${tp}_loop_kernel.json 2> $O/emu.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange fused --tune dec_fused=0 > $O/emu${tp}_loop_fused_3launch.json 2> $O/emu.err
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/tp1_*.json") + glob.glob("$O/emu*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "ms", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], d["config"].get("decode_schedule"), d["config"].get("collective"),
              "gateup us", d["roofline"]["avg_launch_us"], (d.get("emulated_tp") or {}).get("comm_status"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e, open(f.replace(".json", ".err")).read()[-300:] if os.path.exists(f.replace(".json", ".err")) else "")
PY
tail -3 $O/emu.err | cut -c1-300
echo "total $(( $(date +%s) - T0 )) s"
