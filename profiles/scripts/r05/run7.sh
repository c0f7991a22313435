#!/bin/bash
# r05 call 7: is the overlapped schedule held back by the in-region live profiling (cross-stream events on sampled layers) or by the host?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
for cfg in "0 4" "1 4" "1 0" "2 0" "0 0" "1 0" "2 0"; do
  set -- $cfg
  timeout 300 python bench.py --steps 40 --warmup 5 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --tune dec_overlap=$1 --profile-stride $2 > $O/run7_bench_ov$1_ps$2.$RANDOM.json 2> $O/run7_bench.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r05/run7_bench_ov*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["config"].get("decode_schedule"), d["roofline"]["avg_launch_us"], d.get("generate_tokens_per_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
for ov in 0 1 2; do
  timeout 200 python bench.py --steps 40 --warmup 5 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --emulate-tp 8 --tune dec_overlap=$ov --profile-stride 0 > $O/run7_tp8_ov$ov.json 2> $O/run7_tp8_ov$ov.err
  python -c "
import json;d=json.loads(open('$O/run7_tp8_ov$ov.json').read().strip().splitlines()[-1]);print('emulated TP=8 ov=$ov', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['config'].get('decode_schedule'))"
done
