#!/bin/bash
# r06 call 16: flash prefill attention with S one tile ahead (k_attn_fa_pipe, attn_pipe = 1): bit-identity with k_attn_fa, then kernel durations
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
T0=$(date +%s)
(cd $R && timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "variants" > $O/call16_pytest.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s" | tee $O/call16_status.txt)
tail -3 $O/call16_pytest.log | cut -c1-300
B="python3 bench.py --no-cpu-baseline --steps 8 --warmup 2"
for pipe in 0 1; do
  rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- $B --tune attn_pipe=$pipe > $O/c16_tp1_pipe$pipe.json 2>> $O/c16.err)
  python3 $R/profiles/summarize.py "$(find /tmp/kt -name '*.db' | head -1)" 'k_attn_fa' > $O/c16_kstats_tp1_pipe$pipe.txt
  rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- $B --frames 8 --tune attn_pipe=$pipe > $O/c16_frames8_pipe$pipe.json 2>> $O/c16.err)
  python3 $R/profiles/summarize.py "$(find /tmp/kt -name '*.db' | head -1)" 'k_attn_fa' > $O/c16_kstats_frames8_pipe$pipe.txt
done
for f in $O/c16_kstats_*.txt; do echo "== $(basename $f)"; grep -v "^#" $f | head -4 | cut -c1-170; done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c16_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "tok/s", d["value"], "prefill_ms", d.get("prefill_ms"))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
tail -2 $O/c16.err | cut -c1-200
echo "total $(( $(date +%s) - T0 )) s"
