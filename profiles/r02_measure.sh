#!/bin/bash
# Round-2 evidence in one gpurun call (everything lands in gpurun_out/r02/, the judged copies are committed under profiles/):
#   1 default bench line (with the CPU leg)            -> r02_bench_tp1.json
#   2 rocprofv3 kernel trace of the same command       -> r02_kernel_stats_{decode,prefill_encoders}.txt (+ one prefill layer / one encoder pass, dispatch by dispatch)
#   3 PMC passes, each in its own run, --kernel-trace only alongside --pmc (MI355X_MICROARCH.md): FETCH_SIZE, WRITE_SIZE, MFMA busy
#   4 concurrent sequences over the paged KV cache     -> r02_bench_concurrent.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02; mkdir -p $O
(cd $R && timeout 400 python bench.py > $O/r02_bench_tp1.json 2> $O/bench.err)
tail -c 600 $O/r02_bench_tp1.json
rm -rf /tmp/kt; (cd $R && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python bench.py --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.err)
db=$(find /tmp/kt -name '*.db' | head -1)
python $R/profiles/summarize.py $db 'k_dec_' > $O/r02_kernel_stats_decode.txt
python $R/profiles/summarize.py $db 'anonymous namespace' 'k_dec|k_fill_hash' > $O/r02_kernel_stats_prefill_encoders.txt
python $R/profiles/layer_trace.py $db k_moe_sort > $O/r02_prefill_layer_trace.txt
python $R/profiles/layer_trace.py $db k_vit_patchify 3 k_embed_splice > $O/r02_encoder_pass_trace.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  (cd $R && timeout 200 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$ctr -o r -- python bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline > $O/pmc_$ctr.log 2>&1)
  python - "$(find /tmp/pmc_$ctr -name '*.db' | head -1)" $ctr > $O/r02_pmc_$ctr.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("""select name, count(*), avg(counter_value), min(counter_value), max(counter_value), avg(duration)/1e3
                    from pmc_events where counter_name = ? group by name order by 3 desc""", (sys.argv[2],)).fetchall()
print(f"# rocprofv3 --pmc {sys.argv[2]} --kernel-trace -- python bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline")
print(f"# counter {sys.argv[2]} (KiB): name, launch records, mean, min, max, avg_us")
for r in rows[:40]:
    print(f"{r[0][:100]}\t{r[1]}\t{r[2]:.1f}\t{r[3]:.1f}\t{r[4]:.1f}\t{r[5]:.2f}")
PY
done
rm -rf /tmp/pmc_mfma
(cd $R && timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_mfma -o r -- python bench.py --layers 4 --steps 8 --warmup 2 --phase-iters 2 --no-cpu-baseline > $O/pmc_mfma.log 2>&1)
python - "$(find /tmp/pmc_mfma -name '*.db' | head -1)" > $O/r02_pmc_mfma_busy.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, counter_name, count(*), sum(counter_value), avg(duration)/1e3 from pmc_events group by name, counter_name order by name").fetchall()
by = {}
for n, cn, k, v, d in rows:
    by.setdefault(n, {"records": k, "avg_us": d})[cn] = v
print("# MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)")
for n, d in sorted(by.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["records"]):
    if not any(t in n for t in ("k_gemm", "k_attn", "k_dec_")): continue
    busy, mfma = d.get("SQ_BUSY_CU_CYCLES", 0), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    print(f"{n[:80]:80s} avg_us {d['avg_us']:8.1f}  MFMA_BUSY {mfma:14.0f}  BUSY_CU {busy:14.0f}  util {mfma / busy / 4 if busy else 0:.3f}")
PY
(cd $R && timeout 300 python bench.py --no-cpu-baseline --batch 2,3,4,8,16 > $O/r02_bench_concurrent.json 2> $O/conc.err)
python -c "import json,sys; j=json.loads(open('$O/r02_bench_concurrent.json').read().strip().splitlines()[-1]); print(j['value'], j.get('concurrent'))"
head -8 $O/r02_kernel_stats_decode.txt | cut -c1-150; head -14 $O/r02_kernel_stats_prefill_encoders.txt | cut -c1-150
head -12 $O/r02_pmc_mfma_busy.txt | cut -c1-170; head -8 $O/r02_pmc_FETCH_SIZE.txt | cut -c1-170
