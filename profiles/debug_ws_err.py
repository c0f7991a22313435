#!/usr/bin/env python
"""Where is a vh_gemm_ps variant wrong?  gate|up of the MoE micro-benchmark against a torch fp64 reference, error map by
(expert, 16-row tile, 128-column tile).   VITA_AMD_LIB=build/abl/libvita_hip_X.so python profiles/debug_ws_err.py [cfg]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd import _lib, ops
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
S, H, I, E = 552, 4096, 14336, 8
g = torch.Generator(device=dev).manual_seed(0)
W = lambda *s: (torch.randn(s, device=dev, generator=g, dtype=torch.float32) * 0.02).to(torch.bfloat16)
w1, w3 = W(E, I, H), W(E, I, H)
x = torch.randn((S, H), device=dev, generator=g, dtype=torch.float32)
rng = np.random.default_rng(1)
ids = np.stack([rng.permutation(E)[:2] for _ in range(S)]).astype(np.int32)
flat = ids.reshape(-1); order = np.argsort(flat, kind="stable")
rows = np.bincount(flat, minlength=E)
goff = torch.from_numpy(np.concatenate([[0], np.cumsum(rows)]).astype(np.int32)).to(dev)
stok = torch.from_numpy((order // 2).astype(np.int32)).to(dev)
xh, xl = ops.split_planes(x)
_lib.tune("ps_cfg", cfg)
for rep in range(3):
    hh, hl = ops.gemm_ps(xh, xl, w1, w_up=w3, a_rowidx=stok, group_off=goff, ngroups=E, w_group_stride=I * H, m=2 * S, out_split=True)
    torch.cuda.synchronize()
    h = (hh.float() + hl.float())
    bad_total = 0
    for e in range(E):
        r0, r1 = int(goff[e]), int(goff[e + 1])
        xe = x[stok[r0:r1].long()].double()
        gg = xe @ w1[e].double().T; uu = xe @ w3[e].double().T
        ref = (gg / (1 + torch.exp(-gg)) * uu)
        err = torch.nan_to_num((h[r0:r1].double() - ref).abs(), nan=1e30)
        bad = err > 5e-4
        nb = int(bad.sum()); bad_total += nb
        if nb:
            rt = (r1 - r0 + 15) // 16
            tm = torch.zeros(rt, I // 128, dtype=torch.int64)
            bi = bad.nonzero()
            tm.index_put_((bi[:, 0].cpu() // 16, bi[:, 1].cpu() // 128), torch.ones(len(bi), dtype=torch.int64), accumulate=True)
            nz = tm.nonzero()
            print(f"rep {rep} expert {e} rows {r1 - r0}: {nb} bad, nan {int((err > 1e29).sum())}, bad (row tile, n tile) pairs {len(nz)} of {rt * (I // 128)}; "
                  f"row tiles hit {sorted(set(nz[:, 0].tolist()))}; first n tiles {sorted(set(nz[:, 1].tolist()))[:12]}; cols in tile {sorted(set((bi[:, 1].cpu() % 128 // 16).tolist()))}")
    print(f"rep {rep}: total bad {bad_total}")
