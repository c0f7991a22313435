#!/bin/bash
# Round-5 evidence in one gpurun call (everything lands in gpurun_out/r05/, the judged copies are committed under profiles/):
#   1 the driver's exact bench line and the default (with the CPU leg)      -> r05_bench_driver_line.json, r05_bench_tp1.json
#   2 rocprofv3 kernel trace of the default command, both decode schedules  -> r05_kernel_stats_{decode,decode_one_stream,prefill_encoders}.txt,
#                                                                              r05_prefill_layer_trace.txt, r05_encoder_pass_trace.txt
#   3 PMC pass (FETCH_SIZE, own run with --kernel-trace only)               -> r05_pmc_FETCH_SIZE.txt -> r05_pmc_hbm_traffic.json
#   4 one rank's shard at TP = 2 / 4 / 8 (collective skipped)               -> r05_emulated_tp{2,4,8}.json, r05_kernel_stats_tp8.txt
#   5 the N > 1 path on ONE device                                          -> r05_bench_tp8_one_device.json, r05_comm_latency_{2,8}.json
#   6 concurrent sequences                                                  -> r05_bench_concurrent.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
(cd $R && timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r05_bench_driver_line.json 2> $O/bench0.err)
(cd $R && timeout 600 python3 bench.py > $O/r05_bench_tp1.json 2> $O/bench.err)
(cd $R && timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --tune dec_overlap=0 > $O/r05_bench_driver_line_one_stream.json 2> $O/bench1.err)
python3 - <<PY
import json
for f in ("r05_bench_driver_line.json", "r05_bench_tp1.json", "r05_bench_driver_line_one_stream.json"):
    try:
        d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
        print(f, "tok/s", d["value"], d["config"].get("decode_schedule"), "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"], "gen", d.get("generate_tokens_per_s"),
              "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], "rf_prefill", d["roofline_prefill"]["avg_launch_us"], d["roofline_prefill"]["frac"])
    except Exception as e:
        print(f, "no line:", e)
PY
for ov in 1 0; do
  rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python3 bench.py --no-cpu-baseline --tune dec_overlap=$ov > $O/kt_bench_ov$ov.json 2> $O/kt.err)
  db=$(find /tmp/kt -name '*.db' | head -1)
  if [ $ov = 1 ]; then
    python3 $R/profiles/summarize.py $db 'k_dec_' > $O/r05_kernel_stats_decode.txt
    python3 $R/profiles/summarize.py $db 'anonymous namespace' 'k_dec|k_fill_hash' > $O/r05_kernel_stats_prefill_encoders.txt
    python3 $R/profiles/layer_trace.py $db k_moe_sort > $O/r05_prefill_layer_trace.txt
    python3 $R/profiles/layer_trace.py $db k_vit_patchify 2 k_vit_pixel_shuffle > $O/r05_encoder_pass_trace.txt 2>/dev/null
    python3 $R/profiles/layer_trace.py $db "k_dec_gemv<2, 8, true" 400 | cut -c1-140 > $O/r05_decode_layer_timeline.txt
  else
    python3 $R/profiles/summarize.py $db 'k_dec_' > $O/r05_kernel_stats_decode_one_stream.txt
    python3 $R/profiles/layer_trace.py $db "k_dec_gemv<2, 8, true" 400 | cut -c1-140 > $O/r05_decode_layer_timeline_one_stream.txt
  fi
done
rm -rf /tmp/pmc_F
(cd $R && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_F -o r -- python3 bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline > $O/pmc_F.log 2>&1)
python3 - "$(find /tmp/pmc_F -name '*.db' | head -1)" FETCH_SIZE > $O/r05_pmc_FETCH_SIZE.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("""select name, count(*), avg(counter_value), min(counter_value), max(counter_value), avg(duration)/1e3
                    from pmc_events where counter_name = ? group by name order by 3 desc""", (sys.argv[2],)).fetchall()
print(f"# rocprofv3 --pmc {sys.argv[2]} --kernel-trace -- python bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline")
print(f"# counter {sys.argv[2]} (KiB): name, launch records, mean, min, max, avg_us")
for r in rows[:40]:
    print(f"{r[0][:100]}\t{r[1]}\t{r[2]:.1f}\t{r[3]:.1f}\t{r[4]:.1f}\t{r[5]:.2f}")
PY
python3 $R/profiles/make_traffic_json.py $O/r05_pmc_FETCH_SIZE.txt $O/r05_pmc_hbm_traffic.json > /dev/null
for tp in 2 4 8; do
  (cd $R && timeout 300 python3 bench.py --steps 40 --warmup 5 --phase-iters 2 --phase-warmup 1 --no-cpu-baseline --emulate-tp $tp > $O/r05_emulated_tp$tp.json 2> $O/tp.err)
  (cd $R && timeout 300 python3 bench.py --steps 40 --warmup 5 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --emulate-tp $tp --tune dec_overlap=0 > $O/r05_emulated_tp${tp}_one_stream.json 2> $O/tp.err)
done
rm -rf /tmp/kt8; (cd $R && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt8 -o r -- python3 bench.py --steps 24 --warmup 4 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --emulate-tp 8 --tune dec_overlap=0 > $O/kt8.json 2> $O/kt8.err)
python3 $R/profiles/summarize.py $(find /tmp/kt8 -name '*.db' | head -1) 'k_dec_' > $O/r05_kernel_stats_tp8.txt
python3 - <<PY
import json
for tp in (2, 4, 8):
    for sfx in ("", "_one_stream"):
        try:
            d = json.loads(open(f"$O/r05_emulated_tp{tp}{sfx}.json").read().strip().splitlines()[-1])
            print("emulated TP", tp, sfx or "overlapped", d["value"], "tok/s", d["ms_per_step"], "ms", "host", d["host_enqueue_ms_per_step"], "prefill", d["prefill_ms"])
        except Exception as e:
            print(tp, sfx, "no line:", e)
PY
(cd $R && timeout 900 python3 bench.py --gpus 8 --one-device --backend gloo --steps 16 --warmup 4 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline > $O/r05_bench_tp8_one_device.json 2> $O/tp8_one_device.err)
tail -1 $O/r05_bench_tp8_one_device.json | cut -c1-600
for w in 2 8; do (cd $R && timeout 300 python3 profiles/comm_latency.py $w > $O/r05_comm_latency_$w.json 2> $O/comm$w.err); tail -1 $O/r05_comm_latency_$w.json | cut -c1-400; done
(cd $R && timeout 600 python3 bench.py --steps 24 --warmup 4 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --batch 2,3,4,8,16 > $O/r05_bench_concurrent.json 2> $O/conc.err)
head -9 $O/r05_kernel_stats_decode.txt | cut -c1-150; head -9 $O/r05_kernel_stats_decode_one_stream.txt | cut -c1-150; head -14 $O/r05_kernel_stats_prefill_encoders.txt | cut -c1-150
cat $O/r05_decode_layer_timeline.txt; head -10 $O/r05_pmc_FETCH_SIZE.txt | cut -c1-160; head -12 $O/r05_encoder_pass_trace.txt | cut -c1-140
