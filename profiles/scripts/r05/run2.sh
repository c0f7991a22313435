#!/bin/bash
# r05 call 2: first run of the overlapped decode schedule (dec_overlap 0 / 1 / 2): parity, then A/B bench lines at TP = 1 and on one rank's TP = 8 shard
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_mixtral_gpu.py tests/test_fullsize_gpu.py tests/test_comm_gpu.py -m gpu -x -q -s \
    -k "schedules or real_width or overlapped or deterministic or torch_allreduce" 2>&1 | grep -v "^\[stream\]" | tail -25 ) > $O/run2_tests.txt
tail -4 $O/run2_tests.txt | cut -c1-250
for ov in 0 1 2 0 1 2; do
  timeout 300 python bench.py --steps 40 --warmup 5 --phase-iters 2 --phase-warmup 1 --no-cpu-baseline --tune dec_overlap=$ov > $O/run2_bench_ov$ov.$RANDOM.json 2> $O/run2_bench_ov$ov.err
  tail -1 $O/run2_bench_ov$ov.err | cut -c1-200
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r05/run2_bench_ov*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("decode_schedule"), d["roofline"]["avg_launch_us"], d.get("generate_tokens_per_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
for ov in 0 1 2; do
  timeout 200 python bench.py --steps 40 --warmup 5 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --emulate-tp 8 --tune dec_overlap=$ov > $O/run2_tp8_ov$ov.json 2> $O/run2_tp8_ov$ov.err
  python -c "
import json;d=json.loads(open('$O/run2_tp8_ov$ov.json').read().strip().splitlines()[-1]);print('emulated TP=8 ov=$ov', d['value'], d['ms_per_step'], d['config'].get('decode_schedule'))"
done
VITA_REALGEOM_LAYERS=8 timeout 600 python -m pytest tests/test_realgeom_gpu.py -m gpu -x -q -s -k "tp8 or backbone" 2>&1 | grep -v "^\[stream\]" | tail -12 | cut -c1-300 > $O/run2_tp8_8layers.txt
tail -5 $O/run2_tp8_8layers.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pov; timeout 200 rocprofv3 --kernel-trace -d /tmp/pov -o r -- python $R/bench.py --layers 8 --steps 24 --warmup 4 --no-cpu-baseline --phase-iters 1 --phase-warmup 1 --tune dec_overlap=1 > $O/run2_prof_ov1.log 2>&1
python $R/profiles/summarize.py $(find /tmp/pov -name '*.db' | head -1) k_dec | cut -c1-150 | tee $O/run2_kernel_stats_ov1.txt
