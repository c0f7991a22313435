#!/bin/bash
# r04, GPU call 7: flash-form attention (k_attn_fa): parity tests, micro-benchmark A/B, encoder + prefill phases end to end.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run8; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "attention tests rc=$?" | tee -a $O/status.txt; tail -5 $O/pytest_attn.log
timeout 900 python -m pytest tests/test_paged_gpu.py tests/test_mixtral_gpu.py -x -q -k "interleaved or alternative or group4 or real_width" > $O/pytest_engine.log 2>&1; echo "engine tests rc=$?" | tee -a $O/status.txt; tail -5 $O/pytest_engine.log
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
# ---- cost-ordered schedule, XCD mapping inside a round: contiguous chunk per XCD (default build) vs pairs dealt round-robin (map0 build)
for mode in uniform skew; do
  fl=""; [ $mode == skew ] && fl="--skew"
  for rep in 1 2; do
    timeout 300 python profiles/bench_moe_gemm.py --iters 10 $fl > $O/moe_map1_${mode}_$rep.log 2>&1
    VITA_AMD_LIB=$R/build/abl/libvita_hip_sp_map0.so timeout 300 python profiles/bench_moe_gemm.py --iters 10 $fl > $O/moe_map0_${mode}_$rep.log 2>&1
    echo "== $mode rep $rep: map1 (chunk per XCD) / map0 (pairs dealt)"; grep -h "ksplit=auto" $O/moe_map1_${mode}_$rep.log $O/moe_map0_${mode}_$rep.log | cut -c1-170
  done
done | tee $O/moe_map_ab.txt
cd /tmp && export TMPDIR=/tmp
for mp in 1 0; do
  lib=""; [ $mp == 0 ] && lib="$R/build/abl/libvita_hip_sp_map0.so"
  rm -rf /tmp/pmc_map$mp
  (cd $R && VITA_AMD_LIB=$lib timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_map$mp -o r -- python3 bench.py --layers 4 --steps 4 --warmup 2 --phase-iters 2 --no-cpu-baseline > $O/pmc_map$mp.log 2>&1)
  python3 - "$(find /tmp/pmc_map$mp -name '*.db' | head -1)" $mp <<'PY' | tee -a $O/fetch_map_ab.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("""select name, count(*), avg(counter_value), avg(duration)/1e3 from pmc_events where counter_name = 'FETCH_SIZE' and name like '%k_gemm_sp%' group by name""").fetchall()
for n, cnt, v, us in rows:
    print("map%s %-60s launches %4d FETCH_SIZE KiB %.1f -> bytes x2 = %.3f GB  avg %.1f us" % (sys.argv[2], n[:60], cnt, v, v * 1024 * 2 / 1e9, us))
PY
done
