#!/bin/bash
# r06 call 9: gate|up with 7 + 7 rows per block for shards (one round at TP = 8), faster granule polls (second library build): parity, then A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 900 python -m pytest tests/test_comm_gpu.py tests/test_mixtral_gpu.py tests/test_ops_gpu.py -m gpu -x -q --durations=6 > $O/call9_pytest.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call9_status.txt
tail -3 $O/call9_pytest.log | cut -c1-300
B="python3 bench.py --no-cpu-baseline --phase-iters 1 --phase-warmup 1"
for tp in 8 4 2; do
  for rp in 4 7; do
    timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --tune dec_gateup_rp=$rp > $O/c9_emu${tp}_skip_rp$rp.json 2>> $O/c9.err
    timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange fused --tune dec_gateup_rp=$rp > $O/c9_emu${tp}_loop_fused_rp$rp.json 2>> $O/c9.err
  done
  VITA_AMD_LIB=$R/vita_amd/lib/libvita_hip_poll2.so timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange fused > $O/c9_emu${tp}_loop_fused_poll2.json 2>> $O/c9.err
done
timeout 300 $B --steps 40 --warmup 5 > $O/c9_tp1.json 2>> $O/c9.err
VITA_AMD_LIB=$R/vita_amd/lib/libvita_hip_poll2.so timeout 300 $B --steps 40 --warmup 5 > $O/c9_tp1_poll2.json 2>> $O/c9.err
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c9_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "ms", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], d["config"].get("decode_schedule"), d["config"].get("collective"),
              "gateup us", d["roofline"]["avg_launch_us"], (d.get("emulated_tp") or {}).get("comm_status"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
tail -3 $O/c9.err | cut -c1-300
echo "total $(( $(date +%s) - T0 )) s"
