#!/bin/bash
# r05 call 3: overlapped decode schedule with one poller wave per block and per-block sentinels; timeline of two layers
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
( timeout 400 python -m pytest tests/test_mixtral_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -m gpu -x -q \
    -k "schedules or overlapped or vit or tower or encoder" 2>&1 | tail -6 ) > $O/run3_tests.txt
tail -3 $O/run3_tests.txt | cut -c1-250
for ov in 0 2 1 0 2; do
  timeout 300 python bench.py --steps 40 --warmup 5 --phase-iters 2 --phase-warmup 1 --no-cpu-baseline --tune dec_overlap=$ov > $O/run3_bench_ov$ov.$RANDOM.json 2> $O/run3_bench_ov$ov.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r05/run3_bench_ov*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("decode_schedule"), d["roofline"]["avg_launch_us"], d.get("generate_tokens_per_s"), d["vit_projector_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
for ov in 2 0; do
  rm -rf /tmp/pov; timeout 200 rocprofv3 --kernel-trace -d /tmp/pov -o r -- python $R/bench.py --layers 8 --steps 24 --warmup 4 --no-cpu-baseline --phase-iters 1 --phase-warmup 1 --tune dec_overlap=$ov > $O/run3_prof_ov$ov.log 2>&1
  DB=$(find /tmp/pov -name '*.db' | head -1)
  python $R/profiles/summarize.py $DB k_dec | cut -c1-150 > $O/run3_kernel_stats_ov$ov.txt
  python $R/profiles/layer_trace.py $DB "k_dec_gemv<2, 8, true>" 150 | cut -c1-130 > $O/run3_layer_timeline_ov$ov.txt
  python $R/profiles/layer_trace.py $DB "k_dec_gemv<2, 8, true>" 151 | cut -c1-130 >> $O/run3_layer_timeline_ov$ov.txt
done
cat $O/run3_kernel_stats_ov2.txt $O/run3_layer_timeline_ov2.txt
