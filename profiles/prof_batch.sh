#!/bin/bash
# per-kernel table of the batched decode kernels: prof_batch.sh <B> [tune]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-4}; tune=$2
rm -rf /tmp/pb
(cd $R && timeout 200 rocprofv3 --kernel-trace -d /tmp/pb -o r -- python bench.py --layers 8 --steps 16 --warmup 2 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --batch $B ${tune:+--tune $tune} > $R/gpurun_out/pb_$B.log 2>&1)
python $R/profiles/summarize.py $(find /tmp/pb -name '*.db' | head -1) 'k_dec' > $R/gpurun_out/pb_$B.txt
cut -c1-170 $R/gpurun_out/pb_$B.txt
