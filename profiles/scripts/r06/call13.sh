#!/bin/bash
# r06 call 13, the round's closing evidence on the final tree: full GPU suite, smoke, the driver's bench lines, rocprofv3 kernel traces of the
# default command and of the 8-frame shape (-> the r06_kernel_stats_* / r06_kernel_times.json files), FETCH_SIZE of k_attn_fa with / without the
# XCD-aware block order
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
T0=$(date +%s)
(cd $R && timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/call13_pytest_full.log 2>&1; echo "full rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call13_status.txt)
tail -3 $O/call13_pytest_full.log | cut -c1-300
(cd $R && timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/call13_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/call13_status.txt)
(cd $R && timeout 900 python3 bench.py > $O/r06_bench_tp1.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/call13_status.txt)
(cd $R && timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r06_bench_driver_line.json 2> $O/bench0.err)
rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python3 bench.py --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.err)
db=$(find /tmp/kt -name '*.db' | head -1)
python3 $R/profiles/summarize.py $db 'k_dec_' > $O/r06_kernel_stats_decode.txt
python3 $R/profiles/summarize.py $db 'anonymous namespace' 'k_dec|k_fill_hash' > $O/r06_kernel_stats_prefill_encoders.txt
python3 $R/profiles/layer_trace.py $db k_moe_sort > $O/r06_prefill_layer_trace.txt
python3 $R/profiles/layer_trace.py $db k_vit_patchify 2 k_vit_pixel_shuffle > $O/r06_encoder_pass_trace.txt 2>/dev/null
python3 - $O/r06_kernel_stats_decode.txt $O/r06_kernel_stats_prefill_encoders.txt > $O/r06_kernel_times.json <<'PY'
import json, sys
out = {"_how": "rocprofv3 --kernel-trace -- python bench.py --no-cpu-baseline (profiles/scripts/r06/call13.sh); avg_us per kernel of that run"}
for path in sys.argv[1:]:
    for ln in open(path):
        f = ln.split()
        if len(f) < 7 or not f[0].replace(".", "").isdigit():
            continue
        name = " ".join(f[6:])
        for key, pat in (("k_dec_gateup", "k_dec_gateup<2, 4>"), ("k_dec_down", "k_dec_down<7, 2>"), ("k_dec_ablk", "k_dec_ablk<2, 2, 8>"),
                         ("k_dec_lmhead", "k_dec_lmhead<2>"), ("k_gemm_sp_glu", "k_gemm_sp<true"), ("k_attn_fa", "k_attn_fa<1, true, 1>")):
            if pat in name and key not in out:
                out[key] = {"avg_us": float(f[3]), "calls": int(f[1]), "kernel": name[:70]}
print(json.dumps(out, indent=1))
PY
(cd $R && timeout 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --frames 8 > $O/r06_bench_tp1_frames8.json 2> $O/f8.err)
rm -rf /tmp/kt9; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt9 -o r -- python3 bench.py --steps 8 --warmup 2 --phase-iters 3 --no-cpu-baseline --frames 8 > $O/kt9.json 2> $O/kt9.err)
python3 $R/profiles/summarize.py $(find /tmp/kt9 -name '*.db' | head -1) 'anonymous namespace' 'k_dec|k_fill_hash' > $O/r06_kernel_stats_frames8.txt
for x in 1 0; do
  rm -rf /tmp/pmc; (cd $R && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o r -- python3 bench.py --layers 4 --steps 4 --warmup 1 --phase-iters 2 --no-cpu-baseline --frames 8 --tune attn_xcd=$x > $O/pmc_attn_$x.log 2>&1)
  python3 $R/profiles/pmc_table.py "$(find /tmp/pmc -name '*.db' | head -1)" FETCH_SIZE --title "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --layers 4 --steps 4 --warmup 1 --phase-iters 2 --no-cpu-baseline --frames 8 --tune attn_xcd=$x   (raw counter, KiB-units as the guide's HBM section: x 2 correction NOT applied here)" > $O/r06_pmc_attn_fetch_xcd$x.txt
done
cd $R
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/r06_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "ms", d["ms_per_step"], d["config"].get("decode_schedule"),
              "roofline", d["roofline"]["frac"], d["roofline"].get("frac_kernel_trace"), d["roofline"]["avg_launch_us"], "prefill", d["prefill_ms"], "enc", d.get("encode_ms"), "ttft", d["ttft_ms"], d.get("ttft_serial_ms"),
              "rf_prefill", d["roofline_prefill"]["avg_launch_us"], d["roofline_prefill"]["frac"], "gen", d.get("generate_tokens_per_s"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
head -9 $O/r06_kernel_stats_decode.txt | cut -c1-150; head -16 $O/r06_kernel_stats_prefill_encoders.txt | cut -c1-150
head -10 $O/r06_kernel_stats_frames8.txt | cut -c1-150
grep -h "k_attn_fa" $O/r06_pmc_attn_fetch_xcd1.txt $O/r06_pmc_attn_fetch_xcd0.txt | cut -c1-220
echo "total $(( $(date +%s) - T0 )) s"
