#!/bin/bash
# Round-4 evidence in one gpurun call (everything lands in gpurun_out/r04/, the judged copies are committed under profiles/):
#   1 the driver's exact bench line and the default (with the CPU leg)   -> r04_bench_driver_line.json, r04_bench_tp1.json
#   2 rocprofv3 kernel trace of the default command                      -> r04_kernel_stats_{decode,prefill_encoders}.txt, r04_prefill_layer_trace.txt,
#                                                                           r04_encoder_pass_trace.txt
#   3 PMC passes, each in its own run with --kernel-trace only           -> r04_pmc_FETCH_SIZE.txt (-> r04_pmc_hbm_traffic.json), r04_pmc_mfma_busy.txt,
#                                                                           r04_pmc_l2.txt (L2 -> CU requests of the MoE GEMMs)
#   4 video-shaped prompt (configs[4]'s single-GPU shape), duplex hand-off -> r04_bench_tp1_frames8.json, r04_duplex_fullsize.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
(cd $R && timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_driver_line.json 2> $O/bench0.err)
(cd $R && timeout 600 python3 bench.py > $O/r04_bench_tp1.json 2> $O/bench.err)
python3 - <<PY
import json
for f in ("r04_bench_driver_line.json", "r04_bench_tp1.json"):
    try:
        d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
        print(f, "tok/s", d["value"], "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"], "gen", d.get("generate_tokens_per_s"),
              "roofline", d["roofline"]["frac"], "rf_prefill", d["roofline_prefill"]["avg_launch_us"], d["roofline_prefill"]["frac"], "gpu_state", json.dumps(d.get("gpu_state"))[:300])
    except Exception as e:
        print(f, "no line:", e)
PY
rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python3 bench.py --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.err)
db=$(find /tmp/kt -name '*.db' | head -1)
python3 $R/profiles/summarize.py $db 'k_dec_' > $O/r04_kernel_stats_decode.txt
python3 $R/profiles/summarize.py $db 'anonymous namespace' 'k_dec|k_fill_hash' > $O/r04_kernel_stats_prefill_encoders.txt
python3 $R/profiles/layer_trace.py $db k_moe_sort > $O/r04_prefill_layer_trace.txt
python3 $R/profiles/layer_trace.py $db k_vit_patchify 2 k_vit_pixel_shuffle > $O/r04_encoder_pass_trace.txt 2>/dev/null
pmc_tab() {  # db, title
python3 - "$1" "$2" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, counter_name, count(*), avg(counter_value), avg(duration)/1e3 from pmc_events group by name, counter_name order by name").fetchall()
by = {}
for n, cn, k, v, d in rows:
    by.setdefault(n, {"records": k, "avg_us": d})[cn] = v
print("#", sys.argv[2])
for n, d in sorted(by.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["records"])[:24]:
    print(n[:110], "| launch records", d["records"], "| avg_us %.1f" % d["avg_us"])
    for k, v in sorted(d.items()):
        if k not in ("records", "avg_us"): print("    %-32s %16.1f" % (k, v))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("SQ_BUSY_CU_CYCLES"):
        print("    %-32s %16.3f" % ("mfma_busy / (4 x busy_cu)", d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CU_CYCLES"] / 4))
    if "GRBM_GUI_ACTIVE" in d:
        print("    %-32s %16.3f" % ("clock GHz (GUI_ACTIVE / time)", d["GRBM_GUI_ACTIVE"] / d["avg_us"] / 1e3))
PY
}
rm -rf /tmp/pmc_F
(cd $R && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_F -o r -- python3 bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline > $O/pmc_F.log 2>&1)
python3 - "$(find /tmp/pmc_F -name '*.db' | head -1)" FETCH_SIZE > $O/r04_pmc_FETCH_SIZE.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("""select name, count(*), avg(counter_value), min(counter_value), max(counter_value), avg(duration)/1e3
                    from pmc_events where counter_name = ? group by name order by 3 desc""", (sys.argv[2],)).fetchall()
print(f"# rocprofv3 --pmc {sys.argv[2]} --kernel-trace -- python bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline")
print(f"# counter {sys.argv[2]} (KiB): name, launch records, mean, min, max, avg_us")
for r in rows[:40]:
    print(f"{r[0][:100]}\t{r[1]}\t{r[2]:.1f}\t{r[3]:.1f}\t{r[4]:.1f}\t{r[5]:.2f}")
PY
python3 $R/profiles/make_traffic_json.py $O/r04_pmc_FETCH_SIZE.txt $O/r04_pmc_hbm_traffic.json > /dev/null
rm -rf /tmp/pmc_M
(cd $R && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_M -o r -- python3 bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline > $O/pmc_M.log 2>&1)
pmc_tab "$(find /tmp/pmc_M -name '*.db' | head -1)" "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline (per-launch means)" > $O/r04_pmc_mfma_busy.txt
rm -rf /tmp/pmc_L
(cd $R && timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d /tmp/pmc_L -o r -- python3 bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline > $O/pmc_L.log 2>&1)
pmc_tab "$(find /tmp/pmc_L -name '*.db' | head -1)" "rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -- python bench.py --layers 4 ... (per-launch means; L2 <- CU read requests of the streaming GEMMs)" > $O/r04_pmc_l2.txt
(cd $R && timeout 900 python3 bench.py --frames 8 --steps 20 --warmup 5 --phase-iters 3 --no-cpu-baseline > $O/r04_bench_tp1_frames8.json 2> $O/frames8.err)
(cd $R && timeout 900 python3 profiles/duplex_fullsize.py > $O/r04_duplex_fullsize.json 2> $O/duplex.err)
head -9 $O/r04_kernel_stats_decode.txt | cut -c1-150; head -18 $O/r04_kernel_stats_prefill_encoders.txt | cut -c1-150; cat $O/r04_prefill_layer_trace.txt | cut -c1-150
head -8 $O/r04_pmc_FETCH_SIZE.txt | cut -c1-160; head -40 $O/r04_pmc_mfma_busy.txt | cut -c1-150; head -30 $O/r04_pmc_l2.txt | cut -c1-150
tail -3 $O/r04_duplex_fullsize.json | cut -c1-400
