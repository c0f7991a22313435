#!/bin/bash
# MFMA-utilisation PMC pass (own run): SQ_VALU_MFMA_BUSY_CYCLES counts cycles the matrix pipe is busy
# (MI355X_MICROARCH.md: = 32 x N_mfma for 32x32x16 bf16), SQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE give the
# denominator.  Output: gpurun_out/pmc_mfma.txt (per kernel means).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|SQ_BUSY_CU|GRBM_GUI_ACTIVE|SQ_INSTS_VALU " | head -20 > $R/gpurun_out/pmc_list.txt
rm -rf /tmp/pmc_mfma
timeout ${PP_TIMEOUT:-90} rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_mfma -o r -- python $R/bench.py --layers 2 --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_mfma.log 2>&1
db=$(find /tmp/pmc_mfma -name '*.db' | head -1)
python - "$db" > $R/gpurun_out/pmc_mfma.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, counter_name, count(*), sum(counter_value), avg(duration)/1e3 from pmc_events group by name, counter_name order by name").fetchall()
by = {}
for n, cn, k, v, d in rows:
    by.setdefault(n, {"launch_records": k, "avg_us": d})[cn] = v
for n, d in by.items():
    if not any(t in n for t in ("k_gemm", "k_attn", "k_dec_")): continue
    busy, mfma = d.get("SQ_BUSY_CU_CYCLES", 0), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    print(f"{n[:70]:70s} avg_us {d['avg_us']:8.1f}  MFMA_BUSY {mfma:14.0f}  BUSY_CU {busy:14.0f}  WAVE_CYC {d.get('SQ_WAVE_CYCLES',0):14.0f}  "
          f"GUI_ACTIVE {d.get('GRBM_GUI_ACTIVE',0):14.0f}  mfma/busy_cu {mfma / busy if busy else 0:.3f}")
PY
cat $R/gpurun_out/pmc_list.txt | cut -c1-160; cat $R/gpurun_out/pmc_mfma.txt; tail -2 $R/gpurun_out/pmc_mfma.log | cut -c1-200
