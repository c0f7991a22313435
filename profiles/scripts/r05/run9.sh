#!/bin/bash
# r05 call 9: concurrent sequences (B = 2 .. 16) with the batched MoE threshold at 3 (default) and 4; smoke()
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
for bm in 3 4; do
  timeout 400 python bench.py --steps 24 --warmup 4 --phase-iters 1 --phase-warmup 1 --no-cpu-baseline --batch 2,3,4,8,16 --tune batch_moe_min=$bm > $O/run9_concurrent_bm$bm.json 2> $O/run9.err
  python -c "
import json;d=json.loads(open('$O/run9_concurrent_bm$bm.json').read().strip().splitlines()[-1]);print('batch_moe_min=$bm', d['value'], [(c['batch'], c['aggregate_tokens_per_s'], c.get('ms_per_iteration', c.get('ms_per_step'))) for c in d['concurrent']])"
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
