#!/bin/bash
# r04, GPU call 19: which memory-side counters exist (DRAM vs Infinity-Cache), and a pass with them on the MoE GEMM
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run19; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/list_avail.txt 2>&1 || rocprofv3 -L > $O/list_avail.txt 2>&1
grep -i -o "TCC_EA[0-9A-Za-z_]*\|[A-Za-z_]*MALL[A-Za-z_0-9]*\|[A-Za-z_]*DRAM[A-Za-z_0-9]*\|[A-Za-z_]*HBM[A-Za-z_0-9]*" $O/list_avail.txt | sort -u | tr '\n' ' ' | cut -c1-3000
echo
