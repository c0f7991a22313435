#!/bin/bash
# r04, GPU call 24: weight loads of the specialised GEMM without the non-temporal hint (ps_nt = 0): FETCH_SIZE and time of the MoE GEMMs
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run24; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for nt in 0 -1; do
  rm -rf /tmp/pmc_nt$nt
  (cd $R && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_nt$nt -o r -- python3 bench.py --layers 4 --steps 4 --warmup 2 --phase-iters 3 --no-cpu-baseline --tune ps_nt=$nt > $O/pmc_nt$nt.json 2> $O/pmc_nt$nt.err)
  python3 - "$(find /tmp/pmc_nt$nt -name '*.db' | head -1)" $nt <<'PY' | tee -a $O/fetch_nt_ab.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for n, cnt, v, us in c.execute("""select name, count(*), avg(counter_value), avg(duration)/1e3 from pmc_events where counter_name = 'FETCH_SIZE' and name like '%k_gemm_sp%' group by name""").fetchall():
    print("ps_nt=%s %-52s launches %4d FETCH_SIZE KiB %.1f -> x2 = %.3f GB  avg %.1f us" % (sys.argv[2], n[:52], cnt, v, v * 1024 * 2 / 1e9, us))
PY
done
for nt in 0 -1 0 -1; do
  (cd $R && timeout 300 python3 bench.py --layers 8 --steps 4 --warmup 2 --phase-iters 7 --no-cpu-baseline --tune ps_nt=$nt > $O/bench_nt$nt.json 2> $O/bench_nt$nt.err)
  python3 - <<PY | tee -a $O/fetch_nt_ab.txt
import json
d = json.loads(open("$O/bench_nt$nt.json").read().strip().splitlines()[-1])
print("ps_nt=$nt prefill(8 layers) ms", d["prefill_ms"], "min", d["phase_min_ms"]["prefill_ms"], "gate|up live us", d["roofline_prefill"]["avg_launch_us"])
PY
done
