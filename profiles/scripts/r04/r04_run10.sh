#!/bin/bash
# r04, GPU call 9: flash-form attention specialised to d = 128 (8 waves, coalesced K staging), PRE variant removed: parity, A/B,
# FETCH_SIZE of the specialised MoE GEMM under the contiguous-run schedule (PS_SCHED = 0 build) for comparison.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run10; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "attention tests rc=$?" | tee -a $O/status.txt; tail -3 $O/pytest_attn.log
timeout 900 python -m pytest tests/test_paged_gpu.py tests/test_mixtral_gpu.py tests/test_model_gpu.py -x -q > $O/pytest_engine.log 2>&1; echo "engine + model tests rc=$?" | tee -a $O/status.txt; tail -3 $O/pytest_engine.log
timeout 600 python profiles/bench_attn.py --iters 30 --rounds 2 > $O/bench_attn.jsonl 2> $O/bench_attn.err; echo "bench_attn rc=$?" | tee -a $O/status.txt
cat $O/bench_attn.jsonl
for fa in 0 1 0 1; do
  timeout 400 python bench.py --layers 8 --steps 4 --warmup 2 --phase-iters 7 --no-cpu-baseline --tune attn_fa=$fa > $O/bench_fa$fa.json 2> $O/bench_fa$fa.err
  python - <<PY
import json
d = json.loads(open("$O/bench_fa$fa.json").read().strip().splitlines()[-1])
print("attn_fa=$fa", "prefill(8 layers) ms", d["prefill_ms"], "min", d["phase_min_ms"]["prefill_ms"], "vit+proj ms", d["vit_projector_ms"], "min", d["phase_min_ms"]["vit_proj_ms"], "audio", d["audio_encoder_ms"], "tok/s", d["value"])
PY
done | tee $O/phases_ab.txt
cd /tmp && export TMPDIR=/tmp
for sc in 1 0; do
  lib=""; [ $sc == 0 ] && lib="$R/build/abl/libvita_hip_sp_sched0.so"
  for ctr in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $ctr | cut -c1-5); rm -rf /tmp/pmc_s$sc$tag
    (cd $R && VITA_AMD_LIB=$lib timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_s$sc$tag -o r -- python3 bench.py --layers 4 --steps 4 --warmup 2 --phase-iters 2 --no-cpu-baseline > $O/pmc_s$sc$tag.log 2>&1)
    python3 - "$(find /tmp/pmc_s$sc$tag -name '*.db' | head -1)" $sc <<'PY' | tee -a $O/fetch_sched_ab.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("""select name, counter_name, count(*), avg(counter_value), avg(duration)/1e3 from pmc_events where name like '%k_gemm_sp<true%' group by name, counter_name""").fetchall()
for n, cn, cnt, v, us in rows:
    print("PS_SCHED=%s %-44s %-14s launches %4d mean %.1f  avg %.1f us" % (sys.argv[2], n[:44], cn, cnt, v, us))
PY
  done
done
