#!/bin/bash
# r06 call 11: XCD-aware block -> (q tile, KV head) mapping of the attention kernels (attn_xcd) and 32 query rows per wave in the flash prefill
# attention (attn_rows): parity first, then A/B of prefill / encode times and kernel durations (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
T0=$(date +%s)
(cd $R && timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_video_shape_gpu.py tests/test_paged_gpu.py -m gpu -x -q --durations=5 > $O/call11_pytest.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call11_status.txt)
tail -3 $O/call11_pytest.log | cut -c1-300
B="python3 bench.py --no-cpu-baseline --steps 10 --warmup 3"
for arm in "attn_xcd=1" "attn_xcd=0"; do
  (cd $R && timeout 300 $B --tune $arm > $O/c11_tp1_${arm/=/}.json 2>> $O/c11.err)
  for rows in 16 32; do
    (cd $R && timeout 400 $B --frames 8 --tune $arm,attn_rows=$rows > $O/c11_frames8_${arm/=/}_rows$rows.json 2>> $O/c11.err)
  done
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c11_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "prefill_ms", d.get("prefill_ms"), "encode_ms", d.get("encode_ms"), "ttft", d.get("ttft_ms"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
for arm in "attn_xcd=1" "attn_xcd=0"; do
  rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- $B --tune $arm > /dev/null 2>> $O/c11.err)
  python3 $R/profiles/summarize.py "$(find /tmp/kt -name '*.db' | head -1)" 'k_attn|k_rope_kv' > $O/c11_kstats_tp1_${arm/=/}.txt
  for rows in 16 32; do
    rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- $B --frames 8 --tune $arm,attn_rows=$rows > /dev/null 2>> $O/c11.err)
    python3 $R/profiles/summarize.py "$(find /tmp/kt -name '*.db' | head -1)" 'k_attn|k_rope_kv' > $O/c11_kstats_frames8_${arm/=/}_rows$rows.txt
  done
done
for f in $O/c11_kstats_*.txt; do echo "== $(basename $f)"; grep -v "^#" $f | head -8 | cut -c1-200; done
tail -3 $O/c11.err | cut -c1-300
echo "total $(( $(date +%s) - T0 )) s"
