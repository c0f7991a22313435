#!/usr/bin/env python
"""Encoder Linear shapes on the two GEMM kernels: the general 64x128 kernel (vh_gemm: fp32 A split per block, split-K +
reducer for small launches) against the weight-streaming kernel on pre-split planes (vh_gemm_ps; ps_cfg 0 = 64-row
m-tiles, 1 = up to 192-row m-tiles), with the epilogues the encoders use (bias [+ GELU -> planes | + scale + residual]).
   python profiles/bench_enc_gemm.py [--iters 30]        (us per launch, median)"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = [("vit_qkv", 1025, 3072, 1024, "bias"), ("vit_proj", 1025, 1024, 1024, "resid"), ("vit_fc1", 1025, 4096, 1024, "gelu"),
          ("vit_fc2", 1025, 1024, 4096, "resid"), ("vit5_fc1", 5125, 4096, 1024, "gelu"), ("vit5_fc2", 5125, 1024, 4096, "resid"),
          ("aud_qkv", 249, 3072, 1024, "bias"), ("aud_out", 249, 1024, 1024, "resid"), ("aud_w1", 249, 4096, 1024, "gelu"),
          ("aud_w2", 249, 1024, 4096, "resid"), ("proj0", 256, 4096, 4096, "gelu"), ("proj2", 256, 4096, 4096, "bias")]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
    ev[0].record()
    for i in range(args.iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return round(float(np.median([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(args.iters)])), 1)


for name, M, N, K, epi in SHAPES:
    x = torch.randn((M, K), device=dev, generator=g)
    w = (torch.randn((N, K), device=dev, generator=g) * 0.02).to(torch.bfloat16)
    b = torch.randn((N,), device=dev, generator=g)
    sc = torch.randn((N,), device=dev, generator=g)
    res = torch.randn((M, N), device=dev, generator=g)
    out = torch.empty((M, N), device=dev)
    xh, xl = ops.split_planes(x)
    kw = {"bias": dict(bias=b), "gelu": dict(bias=b, act="gelu"), "resid": dict(bias=b, scale=sc, resid=res)}[epi]
    row = {"shape": [M, N, K], "epi": epi}
    row["gemm"] = timeit(lambda: ops.gemm(x, w, out=out, **kw))
    ref = ops.gemm(x, w, **kw)
    for cfg in (0, 1):
        _lib.tune("ps_cfg", cfg)
        if epi == "gelu":
            row[f"ps_cfg{cfg}"] = timeit(lambda: ops.gemm_ps(xh, xl, w, out_split=True, **kw))
            hi, lo = ops.gemm_ps(xh, xl, w, out_split=True, **kw)
            got = hi.float() + lo.float()
        else:
            row[f"ps_cfg{cfg}"] = timeit(lambda: ops.gemm_ps(xh, xl, w, out=out, **kw))
            got = ops.gemm_ps(xh, xl, w, **kw)
        row[f"err{cfg}"] = float((got - ref).abs().max())
    _lib.tune("ps_cfg", -1)
    row["split_planes"] = timeit(lambda: ops.split_planes(x))
    print(json.dumps({name: row}), flush=True)
