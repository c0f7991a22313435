#!/bin/bash
# r04, GPU call 14: dispatch trace of one ViT pass in planes mode (1 tile and 8 frames)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run15; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for fr in 1 8; do
rm -rf /tmp/kt$fr
(cd $R && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt$fr -o r -- python3 bench.py --layers 1 --steps 2 --warmup 1 --phase-iters 2 --no-cpu-baseline --frames $fr > $O/kt$fr.log 2>&1)
db=$(find /tmp/kt$fr -name '*.db' | head -1)
python3 $R/profiles/layer_trace.py $db k_vit_patchify 2 k_vit_pixel_shuffle 2>/dev/null | head -40 | cut -c1-150 > $O/vit_trace_f$fr.txt
cat $O/vit_trace_f$fr.txt | head -26
done
