#!/bin/bash
# r06 call 17: SQ counters of the flash prefill attention at the 8-frame shape (two PMC passes, each its own run with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
T0=$(date +%s)
B="python3 bench.py --layers 4 --steps 4 --warmup 1 --phase-iters 2 --no-cpu-baseline --frames 8"
rm -rf /tmp/pmc; (cd $R && timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmc -o r -- $B > $O/pmc17a.log 2>&1)
python3 $R/profiles/pmc_table.py "$(find /tmp/pmc -name '*.db' | head -1)" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --title "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace -- $B" > $O/r06_pmc_attn_lds.txt
rm -rf /tmp/pmc; (cd $R && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/pmc -o r -- $B > $O/pmc17b.log 2>&1)
python3 $R/profiles/pmc_table.py "$(find /tmp/pmc -name '*.db' | head -1)" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --title "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -- $B" > $O/r06_pmc_attn_sq.txt
grep -h "^#\|k_attn" $O/r06_pmc_attn_lds.txt $O/r06_pmc_attn_sq.txt | cut -c1-400
tail -2 $O/pmc17a.log | cut -c1-200
echo "total $(( $(date +%s) - T0 )) s"
