#!/bin/bash
# r06 call 6: fused MoE launch for shards (k_dec_moe): parity, then one rank's TP = 8 / 4 with and without it; the 8-frame video
# shape at 16 (suite default) and 32 layers with the tie-branch re-run of the oracle
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 900 python -m pytest tests/test_mixtral_gpu.py tests/test_fullsize_gpu.py tests/test_comm_gpu.py tests/test_edge_gpu.py -m gpu -x -q --durations=8 > $O/call6_pytest_a.log 2>&1; echo "pytest a rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call6_status.txt
tail -3 $O/call6_pytest_a.log | cut -c1-300
B="python3 bench.py --no-cpu-baseline --phase-iters 2 --phase-warmup 1"
for tp in 8 4; do
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp > $O/c6_emu${tp}_skip.json 2>> $O/c6.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --tune dec_fused=1 > $O/c6_emu${tp}_skip_nomoe.json 2>> $O/c6.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange fused > $O/c6_emu${tp}_loop_fused.json 2>> $O/c6.err
  timeout 300 $B --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange kernel > $O/c6_emu${tp}_loop_kernel.json 2>> $O/c6.err
done
timeout 300 $B --steps 40 --warmup 5 > $O/c6_tp1.json 2>> $O/c6.err
timeout 300 $B --steps 20 --warmup 5 --frames 8 > $O/c6_tp1_frames8.json 2>> $O/c6.err
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c6_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "ms", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], d["config"].get("decode_schedule"), d["config"].get("collective"),
              "gateup us", d["roofline"]["avg_launch_us"], "prefill", d["prefill_ms"], "enc", d.get("encode_ms"), "ttft", d["ttft_ms"], (d.get("emulated_tp") or {}).get("comm_status"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
timeout 1200 python -m pytest tests/test_video_shape_gpu.py -m gpu -x -q -s > $O/call6_video16.log 2>&1; echo "video16 rc=$? $(( $(date +%s) - T0 )) s" | tee -a $O/call6_status.txt
grep -E "\[video\]|row [0-9]+:|router top-2|hidden after|device ids|logits of|passed|failed" $O/call6_video16.log | cut -c1-250
VITA_VIDEO_LAYERS=32 timeout 2400 python -m pytest tests/test_video_shape_gpu.py -m gpu -x -q -s > $O/r06_video_shape_parity_32.txt 2>&1; echo "video32 rc=$? $(( $(date +%s) - T0 )) s" | tee -a $O/call6_status.txt
grep -E "\[video\]|row [0-9]+:|router top-2|hidden after|device ids|logits of|passed|failed" $O/r06_video_shape_parity_32.txt | cut -c1-250
echo "total $(( $(date +%s) - T0 )) s"
