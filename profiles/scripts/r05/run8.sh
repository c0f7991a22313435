#!/bin/bash
# r05 call 8: gated overlap with gate-based live sampling (no cross-stream events), TP-shard grid rules; default bench flags
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_mixtral_gpu.py tests/test_fullsize_gpu.py tests/test_comm_gpu.py tests/test_ops_gpu.py -m gpu -x -q \
   -k "schedules or overlapped or deterministic or tiny or group4 or world_4_and_8 or tp2_engine or moe_decode or router or lmhead or rope_kv" 2>&1 | tail -5 ) > $O/run8_tests.txt
tail -2 $O/run8_tests.txt | cut -c1-250
for ov in 0 1 0 1; do
  timeout 300 python bench.py --steps 40 --warmup 5 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --tune dec_overlap=$ov > $O/run8_bench_ov$ov.$RANDOM.json 2> $O/run8_bench.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r05/run8_bench_ov*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["config"].get("decode_schedule"), d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d.get("generate_tokens_per_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
for tp in 8 4 2; do for ov in 0 1; do
  timeout 200 python bench.py --steps 40 --warmup 5 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --emulate-tp $tp --tune dec_overlap=$ov --profile-stride 0 > $O/run8_tp${tp}_ov$ov.json 2> $O/run8_tp.err
  python -c "
import json;d=json.loads(open('$O/run8_tp${tp}_ov$ov.json').read().strip().splitlines()[-1]);print('emulated TP=$tp ov=$ov', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['config'].get('decode_schedule'), d['prefill_ms'])"
done; done
