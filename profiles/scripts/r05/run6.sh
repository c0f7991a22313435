#!/bin/bash
# r05 call 6: gated side streams with a gate in front of gate|up (1), without it (2), O gate on QKV start (3), one stream (0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
( timeout 300 python -m pytest tests/test_mixtral_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "schedules or overlapped" 2>&1 | tail -4 ) > $O/run6_tests.txt
tail -2 $O/run6_tests.txt | cut -c1-250
for ov in 0 1 2 3 1 0; do
  timeout 300 python bench.py --steps 40 --warmup 5 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --tune dec_overlap=$ov > $O/run6_bench_ov$ov.$RANDOM.json 2> $O/run6_bench_ov$ov.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r05/run6_bench_ov*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("decode_schedule"), d["roofline"]["avg_launch_us"], d.get("generate_tokens_per_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
for ov in 1; do
  rm -rf /tmp/pov; timeout 200 rocprofv3 --kernel-trace -d /tmp/pov -o r -- python $R/bench.py --layers 8 --steps 24 --warmup 4 --no-cpu-baseline --phase-iters 1 --phase-warmup 1 --tune dec_overlap=$ov > $O/run6_prof_ov$ov.log 2>&1
  DB=$(find /tmp/pov -name '*.db' | head -1)
  python $R/profiles/summarize.py $DB k_dec | cut -c1-150 > $O/run6_kernel_stats_ov$ov.txt
  python $R/profiles/layer_trace.py $DB "k_dec_gemv<2, 8, true" 150 | cut -c1-130 > $O/run6_layer_timeline_ov$ov.txt
  python $R/profiles/layer_trace.py $DB "k_dec_gemv<2, 8, true" 151 | cut -c1-130 >> $O/run6_layer_timeline_ov$ov.txt
  cat $O/run6_kernel_stats_ov$ov.txt $O/run6_layer_timeline_ov$ov.txt
done
