#!/bin/bash
# r05 call 5: gated side streams (dec_overlap 1: O gate on the first QKV granule, 2: both gates on QKV start) against the one-stream schedule
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
( timeout 500 python -m pytest tests/test_mixtral_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -x -q \
    -k "schedules or overlapped or vit or tower or encoder or attention or relpos or flash" 2>&1 | tail -6 ) > $O/run5_tests.txt
tail -3 $O/run5_tests.txt | cut -c1-250
for ov in 0 1 2 0 1 2; do
  timeout 300 python bench.py --steps 40 --warmup 5 --phase-iters 2 --phase-warmup 1 --no-cpu-baseline --tune dec_overlap=$ov > $O/run5_bench_ov$ov.$RANDOM.json 2> $O/run5_bench_ov$ov.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r05/run5_bench_ov*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("decode_schedule"), d["roofline"]["avg_launch_us"], d.get("generate_tokens_per_s"), d["vit_projector_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
for ov in 0 1 2; do
  timeout 200 python bench.py --steps 40 --warmup 5 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --emulate-tp 8 --tune dec_overlap=$ov > $O/run5_tp8_ov$ov.json 2> $O/run5_tp8_ov$ov.err
  python -c "
import json;d=json.loads(open('$O/run5_tp8_ov$ov.json').read().strip().splitlines()[-1]);print('emulated TP=8 ov=$ov', d['value'], d['ms_per_step'], d['config'].get('decode_schedule'))"
done
cd /tmp && export TMPDIR=/tmp
for ov in 1 2; do
  rm -rf /tmp/pov; timeout 200 rocprofv3 --kernel-trace -d /tmp/pov -o r -- python $R/bench.py --layers 8 --steps 24 --warmup 4 --no-cpu-baseline --phase-iters 1 --phase-warmup 1 --tune dec_overlap=$ov > $O/run5_prof_ov$ov.log 2>&1
  DB=$(find /tmp/pov -name '*.db' | head -1)
  python $R/profiles/summarize.py $DB k_dec | cut -c1-150 > $O/run5_kernel_stats_ov$ov.txt
  python $R/profiles/layer_trace.py $DB "k_dec_gemv<2, 8, true" 150 | cut -c1-130 > $O/run5_layer_timeline_ov$ov.txt
  python $R/profiles/layer_trace.py $DB "k_dec_gemv<2, 8, true" 151 | cut -c1-130 >> $O/run5_layer_timeline_ov$ov.txt
  cat $O/run5_kernel_stats_ov$ov.txt $O/run5_layer_timeline_ov$ov.txt
done
