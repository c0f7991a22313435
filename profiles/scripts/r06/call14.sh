#!/bin/bash
# r06 call 14: tile-contiguous copies of the expert weights for the prefill's MoE GEMMs (vh_pack_tiles / moe_tiled): parity, then A/B
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
T0=$(date +%s)
(cd $R && timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_mixtral_gpu.py -m gpu -x -q -k "gemm or tiled or fullsize or real_width or chunked or prefill" --durations=5 > $O/call14_pytest.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call14_status.txt)
tail -3 $O/call14_pytest.log | cut -c1-300
B="python3 bench.py --no-cpu-baseline --steps 10 --warmup 3"
for rep in 1 2; do
  for x in 1 0; do
    (cd $R && timeout 400 $B --tune moe_tiled=$x > $O/c14_tp1_tiled${x}_$rep.json 2>> $O/c14.err)
  done
done
(cd $R && timeout 400 $B --frames 8 --tune moe_tiled=1 > $O/c14_frames8_tiled1.json 2>> $O/c14.err)
(cd $R && timeout 400 $B --frames 8 --tune moe_tiled=0 > $O/c14_frames8_tiled0.json 2>> $O/c14.err)
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c14_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "prefill_ms", d.get("prefill_ms"), "rf_prefill us", d["roofline_prefill"]["avg_launch_us"], "copy", d["config"].get("prefill_weight_copy"), "ttft", d.get("ttft_ms"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
for x in 1 0; do
  rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- $B --tune moe_tiled=$x > /dev/null 2>> $O/c14.err)
  python3 $R/profiles/summarize.py "$(find /tmp/kt -name '*.db' | head -1)" 'k_gemm_sp|k_pack' > $O/c14_kstats_tiled$x.txt
  python3 $R/profiles/layer_trace.py "$(find /tmp/kt -name '*.db' | head -1)" k_moe_sort > $O/c14_layer_tiled$x.txt
done
for f in $O/c14_kstats_*.txt $O/c14_layer_*.txt; do echo "== $(basename $f)"; grep -v "^# rocprofv3" $f | head -9 | cut -c1-170; done
tail -3 $O/c14.err | cut -c1-300
echo "total $(( $(date +%s) - T0 )) s"
