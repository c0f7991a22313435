#!/bin/bash
# r06 call 3: k_dec_ablk with roles ordered by block index (safe under co-tenancy): TP tests on one device, loop-back bit-identity, then A/B:
# gate|up load placement (two library builds), K / V tile images on / off, emulated shards
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests/test_comm_gpu.py -m gpu -x -q --durations=12 > $O/call3_pytest_comm.log 2>&1; echo "pytest comm rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call3_status.txt
tail -3 $O/call3_pytest_comm.log | cut -c1-300
timeout 900 python -m pytest tests/test_mixtral_gpu.py tests/test_fullsize_gpu.py tests/test_duplex_gpu.py tests/test_serving_gpu.py -m gpu -x -q --durations=8 > $O/call3_pytest_b.log 2>&1; echo "pytest b rc=$? $(( $(date +%s) - T0 )) s" | tee -a $O/call3_status.txt
tail -3 $O/call3_pytest_b.log | cut -c1-300
B="python3 bench.py --no-cpu-baseline --phase-iters 3 --phase-warmup 1"
for rep in 1 2; do
  timeout 300 $B --steps 40 --warmup 5 > $O/c3_tp1_r$rep.json 2>> $O/c3.err
  VITA_AMD_LIB=$R/vita_amd/lib/libvita_hip_ge0.so timeout 300 $B --steps 40 --warmup 5 > $O/c3_tp1_ge0_r$rep.json 2>> $O/c3.err
  timeout 300 $B --steps 40 --warmup 5 --tune attn_img=0 > $O/c3_tp1_noimg_r$rep.json 2>> $O/c3.err
done
timeout 300 $B --steps 20 --warmup 5 --frames 8 > $O/c3_tp1_frames8.json 2>> $O/c3.err
timeout 300 $B --steps 20 --warmup 5 --frames 8 --tune attn_img=0 > $O/c3_tp1_frames8_noimg.json 2>> $O/c3.err
for tp in 8 4 2; do
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp > $O/c3_emu${tp}_skip.json 2>> $O/c3.err
  VITA_AMD_LIB=$R/vita_amd/lib/libvita_hip_ge0.so timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp > $O/c3_emu${tp}_skip_ge0.json 2>> $O/c3.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange fused > $O/c3_emu${tp}_loop_fused.json 2>> $O/c3.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange kernel > $O/c3_emu${tp}_loop_kernel.json 2>> $O/c3.err
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c3_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "ms", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], d["config"].get("decode_schedule"), d["config"].get("collective"),
              "gateup us", d["roofline"]["avg_launch_us"], "prefill", d["prefill_ms"], d["phase_min_ms"]["prefill_ms"], "vit", d["vit_projector_ms"], (d.get("emulated_tp") or {}).get("comm_status"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
tail -5 $O/c3.err | cut -c1-300
echo "total $(( $(date +%s) - T0 )) s"
