#!/bin/bash
# r06 call 12: 32 query rows per wave in the flash prefill attention (k_attn_fa<., ., 2>; call 11 ran the library built before that edit):
# parity, then A/B at the 8-frame shape and at configs[2] (rows forced both ways), rocprofv3 kernel durations
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
T0=$(date +%s)
(cd $R && timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_video_shape_gpu.py tests/test_paged_gpu.py -m gpu -x -q -k "attention or variants or video or chunked or paged" --durations=5 > $O/call12_pytest.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call12_status.txt)
tail -3 $O/call12_pytest.log | cut -c1-300
B="python3 bench.py --no-cpu-baseline --steps 10 --warmup 3"
for rows in 16 32; do
  (cd $R && timeout 400 $B --frames 8 --tune attn_rows=$rows > $O/c12_frames8_rows$rows.json 2>> $O/c12.err)
  (cd $R && timeout 400 $B --tune attn_rows=$rows > $O/c12_tp1_rows$rows.json 2>> $O/c12.err)
done
(cd $R && timeout 400 $B --frames 8 > $O/c12_frames8_auto.json 2>> $O/c12.err)
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c12_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "prefill_ms", d.get("prefill_ms"), "encode_ms", d.get("encode_ms"), "ttft", d.get("ttft_ms"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
for rows in 16 32; do
  rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- $B --frames 8 --tune attn_rows=$rows > /dev/null 2>> $O/c12.err)
  python3 $R/profiles/summarize.py "$(find /tmp/kt -name '*.db' | head -1)" 'k_attn|k_rope_kv' > $O/c12_kstats_frames8_rows$rows.txt
  rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- $B --tune attn_rows=$rows > /dev/null 2>> $O/c12.err)
  python3 $R/profiles/summarize.py "$(find /tmp/kt -name '*.db' | head -1)" 'k_attn|k_rope_kv' > $O/c12_kstats_tp1_rows$rows.txt
done
for f in $O/c12_kstats_*.txt; do echo "== $(basename $f)"; grep -v "^#" $f | head -6 | cut -c1-200; done
echo "total $(( $(date +%s) - T0 )) s"
