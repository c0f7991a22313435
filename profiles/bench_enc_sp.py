#!/usr/bin/env python
"""Encoder Linear shapes on the specialised streaming GEMM (vh_gemm_sp, pre-split bf16 planes) with SMALLER m-tiles (rt_cap x 16 rows)
and a K split, against the general 64 x 128 kernel: does a one-round tiling of M ~ 1000 problems beat the register-staged kernel?
   python profiles/bench_enc_sp.py [--iters 20]        (us per launch, median; GEMM only: a split's slabs still need their reducer)"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = [("vit_qkv", 1025, 3072, 1024), ("vit_proj", 1025, 1024, 1024), ("vit_fc1", 1025, 4096, 1024), ("vit_fc2", 1025, 1024, 4096),
          ("vit8_qkv", 8200, 3072, 1024), ("vit8_fc1", 8200, 4096, 1024), ("vit8_fc2", 8200, 1024, 4096),
          ("aud_qkv", 249, 3072, 1024), ("aud_out", 249, 1024, 1024), ("aud_w1", 249, 4096, 1024), ("aud_w2", 249, 1024, 4096)]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
    ev[0].record()
    for i in range(args.iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return round(float(np.median([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(args.iters)])), 1)


nslab = torch.zeros(1, dtype=torch.int32, device=dev)
for name, M, N, K in SHAPES:
    x = torch.randn((M, K), device=dev, generator=g)
    w = (torch.randn((N, K), device=dev, generator=g) * 0.02).to(torch.bfloat16)
    out = torch.empty((M, N), device=dev)
    slabs = torch.empty((8, M, N), device=dev)
    xh, xl = ops.split_planes(x)
    ref = x.double() @ w.double().T
    row = {"shape": [M, N, K], "general": timeit(lambda: ops.gemm(x, w, out=out))}
    for nt in (-1, 0):
        _lib.tune("ps_nt", nt)
        for cap in (0, 3, 4, 5, 6, 8):
            _lib.tune("ps_rtcap", cap)
            for ks in (1, 2, 3, 4, -4):
                if ks != 1 and M * N > 2200 * 4096:
                    continue
                key = f"nt{nt}_cap{cap}_ks{ks}"
                if ks == 1:
                    row[key] = timeit(lambda: ops.gemm_ps(xh, xl, w, out=out))
                    got = ops.gemm_ps(xh, xl, w).double()
                else:
                    row[key] = timeit(lambda: ops.gemm_ps(xh, xl, w, out=slabs[:abs(ks)], ksplit=ks, nslab_out=nslab))
                    ops.gemm_ps(xh, xl, w, out=slabs[:abs(ks)], ksplit=ks, nslab_out=nslab)
                    got = slabs[:(int(nslab.item()) if ks < 0 else ks)].double().sum(0)
                err = float((got - ref).abs().max())
                assert err < 2e-4, (name, key, err)
    _lib.tune("ps_nt", -1); _lib.tune("ps_rtcap", 0)
    best = sorted((v, k) for k, v in row.items() if k.startswith("nt"))[:4]
    print(json.dumps({name: row}), flush=True)
    print(f"## {name:9s} general {row['general']:7.1f}   best streaming: " + "  ".join(f"{k} {v}" for v, k in best), flush=True)
