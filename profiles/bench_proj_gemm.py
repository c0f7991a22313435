#!/usr/bin/env python
"""Micro-benchmark of the prefill projections on the streaming GEMM (vh_gemm_ps): QKV (M = S, N = 6144, K = 4096) and
O (N = 4096) with the device-chosen K split, as vh_api.hip:prefill_impl launches them; prints us per launch and the slab
count.  VITA_AMD_LIB selects an ablated build (profiles/ablate_ps.sh: PS_ABLATE=32 drops the epilogue stores).
   python profiles/bench_proj_gemm.py [--S 552] [--iters 20]"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=552)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--check", action="store_true")
ap.add_argument("--variants", action="store_true", help="also time tile-shape / K-split alternatives (ps_cfg 0 = 64-row m-tiles)")
ap.add_argument("--ab", default="", help="comma list of ps_cfg values to time the two default cases on, interleaved (e.g. 1,2)")
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
S, H = args.S, 4096
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((S, H), device=dev, generator=g)
xh, xl = ops.split_planes(x)
out = {}
nslab = torch.zeros(1, dtype=torch.int32, device=dev)
cases = [("qkv", 6144, -4, -1), ("o", 4096, -8, -1)]
if args.ab:
    cases = [(f"r{r}_{n}_cfg{c}", N, k, int(c)) for r in range(args.rounds) for c in args.ab.split(",") for n, N, k in (("qkv", 6144, -4), ("o", 4096, -8))]
if args.variants:
    cases += [(f"{n}_cfg{c}_ks{k}", N, k, c) for n, N in (("qkv", 6144), ("o", 4096)) for c in (0, 1) for k in (1, 2, 3)]
for name, N, ks, cfg in cases:
    _lib.tune("ps_cfg", cfg)
    ws = [(torch.randn((N, H), device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(3)]
    y = torch.empty((abs(ks), S, N), dtype=torch.float32, device=dev)
    def run(w):
        ops.gemm_ps(xh, xl, w, out=y[0] if ks == 1 else y, ksplit=ks, nslab_out=nslab)
    run(ws[0]); torch.cuda.synchronize()
    if args.check:
        ref = x.double() @ ws[0].double().T
        err = float((y[:int(nslab.item()) if ks < 0 else ks].sum(0).double() - ref).abs().max())
        assert err < 5e-3, err
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
    ev[0].record()
    for i in range(args.iters):
        run(ws[i % 3]); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(args.iters)]
    out[name] = {"us_median": round(float(np.median(ts)), 1), "us_min": round(min(ts), 1), "slabs": int(nslab.item()) if ks < 0 else ks}
    if args.variants or args.ab:
        print(name, out[name], file=sys.stderr, flush=True)
print(json.dumps(out))
