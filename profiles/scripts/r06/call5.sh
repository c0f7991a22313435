#!/bin/bash
# r06 call 5: the whole GPU suite on the current tree (timed against the 1200 s budget), then the headline lines and kernel tables
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > $O/call5_pytest_gpu_full.log 2>&1; echo "full pytest rc=$? in $(( $(date +%s) - T0 )) s" | tee $O/call5_status.txt
grep -E "passed|failed|error" $O/call5_pytest_gpu_full.log | tail -5 | cut -c1-300
B="python3 bench.py --no-cpu-baseline"
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/c5_driver_line.json 2> $O/c5.err
for tp in 8 4 2; do
  timeout 300 $B --phase-iters 2 --phase-warmup 1 --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange fused > $O/c5_emu${tp}_loop_fused.json 2>> $O/c5.err
  timeout 300 $B --phase-iters 2 --phase-warmup 1 --steps 40 --warmup 5 --emulate-tp $tp --loopback --exchange kernel > $O/c5_emu${tp}_loop_kernel.json 2>> $O/c5.err
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python3 bench.py --no-cpu-baseline > $O/c5_kt_bench.json 2> $O/c5_kt.err)
db=$(find /tmp/kt -name '*.db' | head -1)
python3 $R/profiles/summarize.py $db 'k_dec_' > $O/c5_kernel_stats_decode.txt
python3 $R/profiles/summarize.py $db 'anonymous namespace' 'k_dec|k_fill_hash' > $O/c5_kernel_stats_prefill_encoders.txt
python3 $R/profiles/layer_trace.py $db k_moe_sort > $O/c5_prefill_layer_trace.txt
rm -rf /tmp/kt8; (cd $R && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt8 -o r -- python3 bench.py --steps 24 --warmup 4 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --emulate-tp 8 --loopback --exchange fused > $O/c5_kt8.json 2> $O/c5_kt8.err)
python3 $R/profiles/summarize.py $(find /tmp/kt8 -name '*.db' | head -1) 'k_dec_|k_ar_' > $O/c5_kernel_stats_tp8_loop_fused.txt
cd $R
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c5_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "ms", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], d["config"].get("decode_schedule"), d["config"].get("collective"),
              "gateup us", d["roofline"]["avg_launch_us"], "prefill", d["prefill_ms"], "enc", d.get("encode_ms"), "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"], "ttft", d["ttft_ms"], d.get("ttft_serial_ms"), (d.get("emulated_tp") or {}).get("comm_status"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
head -12 $O/c5_kernel_stats_decode.txt | cut -c1-170; head -16 $O/c5_kernel_stats_prefill_encoders.txt | cut -c1-170; head -12 $O/c5_kernel_stats_tp8_loop_fused.txt | cut -c1-170
grep -A26 "slowest" $O/call5_pytest_gpu_full.log | cut -c1-150
echo "total $(( $(date +%s) - T0 )) s"
