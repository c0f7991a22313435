#!/bin/bash
# r03: SQ / LDS counters of the two streaming MoE GEMM kernels (r02 k_gemm_ps, r03 k_gemm_ws) on the micro-benchmark —
# where do the wave-cycles go (parked on s_waitcnt / barrier, issue-stalled, issuing), MFMA pipe busy, LDS activity, clock.
# Counters in their own runs with --kernel-trace only (MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03; mkdir -p $O
pass() {  # name, counters...
  n=$1; shift
  rm -rf /tmp/pmc_$n
  (cd $R && timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$n -o r -- python3 profiles/bench_moe_gemm.py --ab 1,2 --rounds 1 --iters 4 > $O/pmc_$n.log 2>&1)
  python3 - "$(find /tmp/pmc_$n -name '*.db' | head -1)" > $O/r03_pmc_moe_$n.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, counter_name, count(*), avg(counter_value), avg(duration)/1e3 from pmc_events group by name, counter_name order by name").fetchall()
by = {}
for n, cn, k, v, d in rows:
    by.setdefault(n, {"records": k, "avg_us": d})[cn] = v
for n, d in sorted(by.items(), key=lambda kv: -kv[1]["avg_us"]):
    if "k_gemm_" not in n: continue
    print(n[:110], "launches", d["records"], "avg_us %.1f" % d["avg_us"])
    for k, v in sorted(d.items()):
        if k not in ("records", "avg_us"): print("    %-28s %16.0f" % (k, v))
PY
  cat $O/r03_pmc_moe_$n.txt
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
