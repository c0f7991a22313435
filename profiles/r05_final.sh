#!/bin/bash
# r05, final evidence on the final tree: the whole GPU suite (what the driver runs, < 1200 s), then the headline lines and decode kernel tables
# (the rest of the round's measurements — emulated shards, N > 1 on one device, PMC, concurrency — is profiles/r05_measure.sh, run on the tree of
# commit "r05 measurement scripts"; kernels unchanged since)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/r05_pytest_gpu_full.log 2>&1; echo "full pytest rc=$? in $(( $(date +%s) - T0 )) s" | tee $O/status.txt
grep -E "passed|failed|error|\[realgeom\] TP" $O/r05_pytest_gpu_full.log | tail -8 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r05_bench_driver_line.json 2> $O/bench0.err)
(cd $R && timeout 600 python3 bench.py > $O/r05_bench_tp1.json 2> $O/bench.err)
(cd $R && timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --tune dec_overlap=1 > $O/r05_bench_driver_line_overlapped.json 2> $O/bench1.err)
python3 - <<PY
import json
for f in ("r05_bench_driver_line.json", "r05_bench_tp1.json", "r05_bench_driver_line_overlapped.json"):
    try:
        d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
        print(f, "tok/s", d["value"], d["config"].get("decode_schedule"), "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"], "gen", d.get("generate_tokens_per_s"),
              "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["traffic"], "rf_prefill", d["roofline_prefill"]["avg_launch_us"], d["roofline_prefill"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "no line:", e)
PY
for ov in -1 1; do
  rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python3 bench.py --no-cpu-baseline --tune dec_overlap=$ov > $O/kt_bench_ov$ov.json 2> $O/kt.err)
  db=$(find /tmp/kt -name '*.db' | head -1)
  if [ $ov = -1 ]; then
    python3 $R/profiles/summarize.py $db 'k_dec_' > $O/r05_kernel_stats_decode.txt
    python3 $R/profiles/summarize.py $db 'anonymous namespace' 'k_dec|k_fill_hash' > $O/r05_kernel_stats_prefill_encoders.txt
    python3 $R/profiles/layer_trace.py $db k_moe_sort > $O/r05_prefill_layer_trace.txt
    python3 $R/profiles/layer_trace.py $db k_vit_patchify 2 k_vit_pixel_shuffle > $O/r05_encoder_pass_trace.txt 2>/dev/null
    python3 $R/profiles/layer_trace.py $db "k_dec_gemv<2, 8, true" 400 | cut -c1-140 > $O/r05_decode_layer_timeline.txt
  else
    python3 $R/profiles/summarize.py $db 'k_dec_' > $O/r05_kernel_stats_decode_overlapped.txt
    python3 $R/profiles/layer_trace.py $db "k_dec_gemv<2, 8, true" 400 | cut -c1-140 > $O/r05_decode_layer_timeline_overlapped.txt
  fi
done
head -9 $O/r05_kernel_stats_decode.txt | cut -c1-150; head -10 $O/r05_kernel_stats_decode_overlapped.txt | cut -c1-150; head -12 $O/r05_kernel_stats_prefill_encoders.txt | cut -c1-150
cat $O/r05_decode_layer_timeline_overlapped.txt; head -12 $O/r05_encoder_pass_trace.txt | cut -c1-140
