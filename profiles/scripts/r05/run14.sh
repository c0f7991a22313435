#!/bin/bash
# r05 call 14: (1) concurrency with r02's own command line (--steps 64 --warmup 8: r05_bench_concurrent.json used 24 / 4, i.e. other decode
# trajectories and other touched-expert counts per iteration), with the B = 3 crossover A/B; (2) config 5's single-GPU shape (8 frames, S = 2344)
# on the final tree + its kernel table.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 400 python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --phase-iters 1 --batch 2,3,4,8,16 > $O/r05_bench_concurrent_k64.json 2> $O/c64.err)
(cd $R && timeout 300 python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --phase-iters 1 --batch 3,4 --tune batch_moe_min=5 > $O/run14_concurrent_k64_gemv.json 2> $O/c64g.err)
(cd $R && timeout 400 python3 bench.py --frames 8 --steps 32 --warmup 8 --no-cpu-baseline > $O/r05_bench_tp1_frames8.json 2> $O/f8.err)
python3 - <<PY
import json
for f in ("r05_bench_concurrent_k64.json", "run14_concurrent_k64_gemv.json", "r05_bench_tp1_frames8.json"):
    try:
        d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
        print(f, "tok/s", d["value"], "prefill", d["prefill_ms"], "vit", d["vit_projector_ms"], "aud", d["audio_encoder_ms"], "S", d["config"].get("prompt_tokens"))
        for c in d.get("concurrent", []): print("   B", c["batch"], c["aggregate_tokens_per_s"], c["ms_per_iteration"])
    except Exception as e:
        print(f, "no line:", e)
PY
rm -rf /tmp/kt; (cd $R && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python3 bench.py --frames 8 --steps 8 --warmup 2 --no-cpu-baseline --phase-iters 2 > $O/kt_f8.json 2> $O/kt_f8.err)
db=$(find /tmp/kt -name '*.db' | head -1)
python3 $R/profiles/summarize.py $db 'anonymous namespace' 'k_dec|k_fill_hash' > $O/r05_kernel_stats_frames8.txt
head -16 $O/r05_kernel_stats_frames8.txt | cut -c1-160
