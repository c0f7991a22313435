#!/usr/bin/env python
"""Attention kernel micro-benchmark at the hot shapes: ViT (16 heads x 64, N = 1025, no mask; 1 and 8 images per launch) and the
Mixtral prefill (32 q / 8 kv heads x 128, causal, KV-cache layout; S = 552 and the 8-frame video prompt's S = 2344), over the
kernel variants (vh_tune attn_fa / attn_rows / attn_ksplit / attn_impl), interleaved in ONE process, --rounds times.  us per
launch (median of --iters, launches queued back to back) and the max |error| of each variant against an fp64 torch reference.
   python profiles/bench_attn.py [--iters 30] [--rounds 2]"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--only-default", action="store_true", help="ablated builds (VITA_AMD_LIB): time the default variant only, no reference check")
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


def ref_attn(q, k, v, scale, causal):          # q [H, Sq, d], k / v [Hkv, Sk, d] (fp64 on the device)
    H, Sq, d = q.shape
    grp = H // k.shape[0]
    s = torch.einsum("hqd,hkd->hqk", q.double(), k.double().repeat_interleave(grp, 0)) * scale
    if causal:
        s = s.masked_fill(torch.ones(Sq, k.shape[1], dtype=torch.bool, device=q.device).triu(1 + k.shape[1] - Sq), float("-inf"))
    return torch.einsum("hqk,hkd->qhd", torch.softmax(s, -1), v.double().repeat_interleave(grp, 0)).reshape(Sq, H * d)


cases = {}
H, N, d = 16, 1025, 64
C = H * d
for nimg in (1, 8):
    qkv = torch.randn((nimg * N, 3 * C), device=dev, generator=g)
    out = torch.empty((nimg * N, C), device=dev)
    fn = (lambda qkv=qkv, out=out, nimg=nimg: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], out, B=nimg, Hq=H, Hkv=H, Sq=N, Sk=N, d=d,
                                                             ldq=3 * C, hsq=d, ldk=3 * C, hsk=d, ldv=3 * C, hsv=d, ldo=C, bsq=N * 3 * C,
                                                             bsk=N * 3 * C, bso=N * C, scale=d ** -0.5))
    r = qkv[:N].reshape(N, 3, H, d).permute(1, 2, 0, 3)
    cases[f"vit_x{nimg}"] = (fn, out, ref_attn(r[0], r[1], r[2], d ** -0.5, False), N)
nq, nkv, d2 = 32, 8, 128
for S in (552, 2344):
    ctx = S + 88
    q2 = torch.randn((S, nq * d2), device=dev, generator=g)
    kc = torch.randn((nkv, ctx, d2), device=dev, generator=g)
    vc = torch.randn((nkv, ctx, d2), device=dev, generator=g)
    out2 = torch.empty((S, nq * d2), device=dev)
    fn = (lambda q2=q2, kc=kc, vc=vc, out2=out2, S=S, ctx=ctx: ops.attention(q2, kc, vc, out2, B=1, Hq=nq, Hkv=nkv, Sq=S, Sk=S, d=d2,
                                                                              ldq=nq * d2, hsq=d2, ldk=d2, hsk=ctx * d2, ldv=d2, hsv=ctx * d2,
                                                                              ldo=nq * d2, scale=d2 ** -0.5, causal=True, q_off=0))
    cases[f"prefill_S{S}"] = (fn, out2, ref_attn(q2.reshape(S, nq, d2).permute(1, 0, 2), kc[:, :S], vc[:, :S], d2 ** -0.5, True), S)

DEFAULTS = {"attn_impl": 0, "attn_ksplit": 0, "attn_rows": 0, "attn_fa": 1}
VARIANTS = [{}, {"attn_fa": 0}, {"attn_fa": 0, "attn_rows": 16}, {"attn_fa": 0, "attn_ksplit": 2}, {"attn_impl": 2}]
if args.only_default:
    VARIANTS = [{}]
for rnd in range(args.rounds):
    for v in VARIANTS:
        for k, val in {**DEFAULTS, **v}.items():
            _lib.tune(k, val)
        row = {"round": rnd, "variant": v or "default"}
        for name, (fn, out, ref, rows) in cases.items():
            row[name + "_us"] = timeit(fn)
            if rnd == 0 and not args.only_default:
                out.fill_(float("nan"))
                fn()
                row[name + "_err"] = float("%.2e" % float(torch.nan_to_num((out[:rows].double() - ref).abs(), nan=1e30).max()))
        print(json.dumps(row), flush=True)
for k, val in DEFAULTS.items():
    _lib.tune(k, val)
