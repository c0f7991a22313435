#!/bin/bash
# Round-3 evidence in one gpurun call (everything lands in gpurun_out/r03/, the judged copies are committed under profiles/):
#   1 the driver's exact bench line and the default (with the CPU leg)   -> r03_bench_driver_line.json, r03_bench_tp1.json
#   2 rocprofv3 kernel trace of the default command                      -> r03_kernel_stats_{decode,prefill_encoders}.txt (+ one prefill layer)
#   3 PMC FETCH_SIZE in its own run, --kernel-trace only alongside --pmc -> r03_pmc_FETCH_SIZE.txt, r03_pmc_hbm_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03; mkdir -p $O
(cd $R && timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_bench_driver_line.json 2> $O/bench0.err)
(cd $R && timeout 500 python3 bench.py > $O/r03_bench_tp1.json 2> $O/bench.err)
python3 - <<PY
import json
for f in ("r03_bench_driver_line.json", "r03_bench_tp1.json"):
    d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
    print(f, "tok/s", d["value"], "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"], "gen", d.get("generate_tokens_per_s"),
          "roofline", d["roofline"]["frac"], "gpu_state", json.dumps(d.get("gpu_state"))[:400])
PY
rm -rf /tmp/kt; (cd $R && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python3 bench.py --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.err)
db=$(find /tmp/kt -name '*.db' | head -1)
python3 $R/profiles/summarize.py $db 'k_dec_' > $O/r03_kernel_stats_decode.txt
python3 $R/profiles/summarize.py $db 'anonymous namespace' 'k_dec|k_fill_hash' > $O/r03_kernel_stats_prefill_encoders.txt
python3 $R/profiles/layer_trace.py $db k_moe_sort > $O/r03_prefill_layer_trace.txt
rm -rf /tmp/pmc_F
(cd $R && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_F -o r -- python3 bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline > $O/pmc_F.log 2>&1)
python3 - "$(find /tmp/pmc_F -name '*.db' | head -1)" FETCH_SIZE > $O/r03_pmc_FETCH_SIZE.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("""select name, count(*), avg(counter_value), min(counter_value), max(counter_value), avg(duration)/1e3
                    from pmc_events where counter_name = ? group by name order by 3 desc""", (sys.argv[2],)).fetchall()
print(f"# rocprofv3 --pmc {sys.argv[2]} --kernel-trace -- python bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline")
print(f"# counter {sys.argv[2]} (KiB): name, launch records, mean, min, max, avg_us")
for r in rows[:40]:
    print(f"{r[0][:100]}\t{r[1]}\t{r[2]:.1f}\t{r[3]:.1f}\t{r[4]:.1f}\t{r[5]:.2f}")
PY
head -8 $O/r03_kernel_stats_decode.txt | cut -c1-150; head -16 $O/r03_kernel_stats_prefill_encoders.txt | cut -c1-150; head -8 $O/r03_pmc_FETCH_SIZE.txt | cut -c1-160
