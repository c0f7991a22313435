#!/usr/bin/env python
"""Attention kernel micro-benchmark at the two hot shapes: ViT (1 tile: 16 heads x 64, N = 1025, no mask) and the Mixtral
prefill (32 q / 8 kv heads x 128, S = 552, causal, KV-cache layout), over kernel variants (vh_tune attn_impl / attn_ksplit /
attn_wpe / attn_rows).  us per launch (median of --iters, launches queued back to back).
   python profiles/bench_attn.py [--iters 30]"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
    ev[0].record()
    for i in range(args.iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return round(float(np.median([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(args.iters)])), 1)


H, N, d = 16, 1025, 64
C = H * d
qkv = torch.randn((N, 3 * C), device=dev, generator=g)
out = torch.empty((N, C), device=dev)
ws1 = torch.empty(ops.attention_ws_bytes(1, H, N, d), dtype=torch.uint8, device=dev)
vit = lambda: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], out, B=1, Hq=H, Hkv=H, Sq=N, Sk=N, d=d, ldq=3 * C, hsq=d, ldk=3 * C,
                            hsk=d, ldv=3 * C, hsv=d, ldo=C, scale=d ** -0.5, ws=ws1)
nq, nkv, d2, S, ctx = 32, 8, 128, 552, 640
q2 = torch.randn((S, nq * d2), device=dev, generator=g)
kc = torch.randn((nkv, ctx, d2), device=dev, generator=g)
vc = torch.randn((nkv, ctx, d2), device=dev, generator=g)
out2 = torch.empty((S, nq * d2), device=dev)
ws2 = torch.empty(ops.attention_ws_bytes(1, nkv, S, d2), dtype=torch.uint8, device=dev)
pre = lambda: ops.attention(q2, kc, vc, out2, B=1, Hq=nq, Hkv=nkv, Sq=S, Sk=S, d=d2, ldq=nq * d2, hsq=d2, ldk=d2, hsk=ctx * d2,
                            ldv=d2, hsv=ctx * d2, ldo=nq * d2, scale=d2 ** -0.5, causal=True, q_off=0, ws=ws2)
DEFAULTS = {"attn_impl": 0, "attn_ksplit": 0, "attn_wpe": 0, "attn_rows": 0, "attn_presplit": 0}
VARIANTS = [{}, {"attn_presplit": 1}, {"attn_rows": 16}, {"attn_rows": 16, "attn_presplit": 1}, {"attn_impl": 2}, {"attn_ksplit": 1}, {"attn_ksplit": 2}, {"attn_wpe": 3}, {"attn_ksplit": 2, "attn_wpe": 3},
            {"attn_rows": 32}, {"attn_rows": 32, "attn_ksplit": 2}, {"attn_rows": 32, "attn_ksplit": 1}]
for v in VARIANTS:
    for k, val in {**DEFAULTS, **v}.items():
        _lib.tune(k, val)
    print(json.dumps({"variant": v or "default", "vit_us": timeit(vit), "prefill_us": timeit(pre)}), flush=True)
for k, val in DEFAULTS.items():
    _lib.tune(k, val)
