#!/bin/bash
# r06 call 2: exchange polls in parallel, balanced gate|up grid for shards, K / V tile images for the flash prefill attention:
# parity first (TP tests, loop-back bit-identity, prefill parity at real width), then the A/B numbers
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 1200 python -m pytest tests/test_comm_gpu.py -m gpu -x -q --durations=12 > $O/call2_pytest_comm.log 2>&1; echo "pytest comm rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call2_status.txt
tail -3 $O/call2_pytest_comm.log | cut -c1-300
timeout 1200 python -m pytest tests/test_mixtral_gpu.py tests/test_fullsize_gpu.py tests/test_ops_gpu.py tests/test_paged_gpu.py tests/test_model_gpu.py -m gpu -x -q --durations=12 > $O/call2_pytest_b.log 2>&1; echo "pytest b rc=$? $(( $(date +%s) - T0 )) s" | tee -a $O/call2_status.txt
tail -3 $O/call2_pytest_b.log | cut -c1-300
B="python3 bench.py --no-cpu-baseline --phase-iters 3 --phase-warmup 1"
timeout 300 $B --steps 40 --warmup 5 > $O/c2_tp1.json 2> $O/c2.err
timeout 300 $B --steps 20 --warmup 5 --frames 8 > $O/c2_tp1_frames8.json 2>> $O/c2.err
timeout 300 $B --steps 20 --warmup 5 --frames 8 --tune attn_fa=0 > $O/c2_tp1_frames8_nofa.json 2>> $O/c2.err
for tp in 8 4 2; do
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp > $O/c2_emu${tp}_skip.json 2>> $O/c2.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange fused > $O/c2_emu${tp}_loop_fused.json 2>> $O/c2.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange kernel > $O/c2_emu${tp}_loop_kernel.json 2>> $O/c2.err
done
for g in 384 512 640 896; do
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp 8 --tune dec_gateup_grid=$g > $O/c2_emu8_skip_grid$g.json 2>> $O/c2.err
done
for g in 384 512; do
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp 4 --tune dec_gateup_grid=$g > $O/c2_emu4_skip_grid$g.json 2>> $O/c2.err
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c2_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "ms", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], d["config"].get("decode_schedule"), d["config"].get("collective"),
              "gateup us", d["roofline"]["avg_launch_us"], "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], (d.get("emulated_tp") or {}).get("comm_status"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
tail -5 $O/c2.err | cut -c1-300
echo "total $(( $(date +%s) - T0 )) s"
