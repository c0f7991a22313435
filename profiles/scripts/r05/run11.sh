#!/bin/bash
# r05 call 11: video-shaped prompt at 16 layers; bench sanity with the sampling stride at 8; shard-shape grid sweep (experiment libraries built with
# -DVH_EXP_GATEUP_GRID=n / -DVH_EXP_QKV_R=n from a local edit of vh_decode.hip, selected through VITA_AMD_LIB; not committed)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_video_shape_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^\[stream\]" | tail -8 ) > $O/run11_video16.txt
tail -3 $O/run11_video16.txt | cut -c1-250
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/run11_bench_driver_line.json 2> $O/run11.err
python -c "
import json;d=json.loads(open('$O/run11_bench_driver_line.json').read().strip().splitlines()[-1]);print('driver line', d['value'], d['ms_per_step'], d['config']['decode_schedule'], d['roofline']['frac'], d['roofline']['samples'], d['prefill_ms'], d['vit_projector_ms'])"
for lib in default GATEUP_GRID_448 GATEUP_GRID_512 GATEUP_GRID_640 GATEUP_GRID_896 QKV_R_4 QKV_R_2; do
  for tp in 8 4; do
    if [ $lib = default ]; then unset VITA_AMD_LIB; else export VITA_AMD_LIB=$R/build/exp/libvita_hip_$lib.so; fi
    timeout 200 python bench.py --steps 40 --warmup 5 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --emulate-tp $tp --profile-stride 0 > $O/run11_tp${tp}_$lib.json 2> $O/run11_tp.err
    python -c "
import json;d=json.loads(open('$O/run11_tp${tp}_$lib.json').read().strip().splitlines()[-1]);print('emulated TP=$tp $lib', d['value'], d['ms_per_step'], d['config'].get('decode_schedule'))"
  done
done
unset VITA_AMD_LIB
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
