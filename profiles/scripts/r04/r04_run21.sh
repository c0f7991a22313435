#!/bin/bash
# r04, GPU call 21: concurrent sequences over the paged KV cache on the round's kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python bench.py --batch 2,3,4,8,16 --no-cpu-baseline > $O/r04_bench_concurrent.json 2> $O/concurrent.err; echo "rc=$?"
python - <<PY
import json
d = json.loads(open("$O/r04_bench_concurrent.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("concurrent", d.get("batch", {})))[:1500]); print("tok/s", d["value"])
PY
