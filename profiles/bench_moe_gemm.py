#!/usr/bin/env python
"""Micro-benchmark of the prefill MoE GEMM pair at the released geometry (one layer: 8 experts, H 4096, I 14336),
S tokens with random top-2 routing.  Times gate|up and down per launch with events on the launch stream, checks a
sample of the output against an fp64 torch reference, and sweeps the kernel's tuning knobs.

  python profiles/bench_moe_gemm.py [--S 552] [--iters 10] [--sweep]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=552)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--nocheck", action="store_true", help="ablated builds: skip the parity assertion")
    ap.add_argument("--skew", action="store_true", help="imbalanced routing: expert probabilities of a real layer "
                                                        "(76 .. 307 of 1136 rows) instead of uniform")
    ap.add_argument("--H", type=int, default=4096)
    ap.add_argument("--I", type=int, default=14336)
    ap.add_argument("--E", type=int, default=8)
    ap.add_argument("--ab", default="", help="comma list of ps_cfg values to A/B in interleaved rounds, e.g. 1,2")
    ap.add_argument("--abtune", default="", help="interleaved A/B of one vh_tune key over values, e.g. ps_xcd=0,1,3 (r06); parity of every value first")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--layers", type=int, default=2, help="distinct weight sets cycled through (defeats cache reuse)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    S, H, I, E = args.S, args.H, args.I, args.E
    g = torch.Generator(device=dev).manual_seed(0)

    def W(*shape):
        return (torch.randn(shape, device=dev, generator=g, dtype=torch.float32) * 0.02).to(torch.bfloat16)

    layers = [dict(w1=W(E, I, H), w3=W(E, I, H), w2=W(E, H, I)) for _ in range(args.layers)]
    x = torch.randn((S, H), device=dev, generator=g, dtype=torch.float32)
    rng = np.random.default_rng(1)
    if args.skew:
        pr = np.asarray([107, 114, 89, 76, 82, 307, 140, 221], np.float64)[:E]
        pr = pr / pr.sum()
        ids = np.stack([rng.choice(E, size=2, replace=False, p=pr) for _ in range(S)]).astype(np.int32)
    else:
        ids = np.stack([rng.permutation(E)[:2] for _ in range(S)]).astype(np.int32)
    flat = ids.reshape(-1)
    order = np.argsort(flat, kind="stable")
    goff = torch.from_numpy(np.concatenate([[0], np.cumsum(np.bincount(flat, minlength=E))]).astype(np.int32)).to(dev)
    stok = torch.from_numpy((order // 2).astype(np.int32)).to(dev)
    sslot = torch.from_numpy(order.astype(np.int32)).to(dev)
    xh, xl = ops.split_planes(x)
    rows = np.bincount(flat, minlength=E)
    print("rows per expert:", rows.tolist(), flush=True)

    nslab = torch.zeros(1, dtype=torch.int32, device=dev)

    def run_ps(L, ksplit):
        hh, hl = ops.gemm_ps(xh, xl, L["w1"], w_up=L["w3"], a_rowidx=stok, group_off=goff, ngroups=E,
                             w_group_stride=I * H, m=2 * S, out_split=True)
        y = torch.zeros((abs(ksplit), 2 * S, H), dtype=torch.float32, device=dev)
        ops.gemm_ps(hh, hl, L["w2"], group_off=goff, ngroups=E, w_group_stride=H * I, c_rowidx=sslot,
                    out=y if ksplit != 1 else y[0], ksplit=ksplit, nslab_out=nslab)
        if ksplit < 0:
            y = y[:int(nslab.item())]
        return hh, hl, y

    def run_general(L):
        h = ops.gemm(x, L["w1"], w_up=L["w3"], a_rowidx=stok, group_off=goff, ngroups=E, w_group_stride=I * H, m=2 * S)
        y = torch.empty((2 * S, H), dtype=torch.float32, device=dev)
        ops.gemm(h, L["w2"], group_off=goff, ngroups=E, w_group_stride=H * I, c_rowidx=sslot, out=y)
        return h, y

    # ---- correctness on a sample (fp64 torch reference) ------------------------------------------------
    L = layers[0]
    hh, hl, y = run_ps(L, -4)
    torch.cuda.synchronize()
    print("device-chosen K split of the down projection:", int(nslab.item()), flush=True)
    h = hh.float() + hl.float()
    ysum = y.sum(0)
    err_h = err_y = 0.0
    for p in list(range(0, 2 * S, max(1, (2 * S) // 24))) + [2 * S - 1]:
        slot = int(order[p]); e = int(flat[slot]); t = slot // 2
        xr = x[t].double()
        gg = L["w1"][e].double() @ xr
        uu = L["w3"][e].double() @ xr
        href = gg / (1 + torch.exp(-gg)) * uu
        err_h = max(err_h, float(torch.nan_to_num((h[p].double() - href).abs(), nan=1e30).max()))
        yref = L["w2"][e].double() @ h[p].double()
        err_y = max(err_y, float(torch.nan_to_num((ysum[slot].double() - yref).abs(), nan=1e30).max()))
    print(f"max |err| gate/up {err_h:.3e}   down {err_y:.3e}", flush=True)
    assert args.nocheck or (err_h < 5e-4 and err_y < 5e-4), "parity failure"

    def timed(fn, iters):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        for i in range(2):
            fn(layers[i % len(layers)])
        torch.cuda.synchronize()
        ev[0].record()
        for i in range(iters):
            fn(layers[i % len(layers)])
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(iters)]
        return float(np.median(ts)), float(np.min(ts))

    def time_pair(ksplit):
        hh, hl, _ = run_ps(layers[0], ksplit)
        y = torch.empty((abs(ksplit), 2 * S, H), dtype=torch.float32, device=dev)

        def gu(L):
            ops.gemm_ps(xh, xl, L["w1"], w_up=L["w3"], a_rowidx=stok, group_off=goff, ngroups=E,
                        w_group_stride=I * H, m=2 * S, out_split=True)

        def dn(L):
            ops.gemm_ps(hh, hl, L["w2"], group_off=goff, ngroups=E, w_group_stride=H * I, c_rowidx=sslot,
                        out=y if ksplit != 1 else y[0], ksplit=ksplit, nslab_out=nslab)

        return timed(gu, args.iters), timed(dn, args.iters)

    out = {"S": S, "rows": rows.tolist()}
    gu_bytes = 2 * E * I * H * 2
    dn_bytes = E * I * H * 2

    def report(tag, gu, dn):
        print(f"{tag:34s} gate|up {gu[0]:8.1f} us (min {gu[1]:8.1f}) = {gu_bytes / gu[0] / 1e6:5.2f} TB/s | "
              f"down {dn[0]:8.1f} us (min {dn[1]:8.1f}) = {dn_bytes / dn[0] / 1e6:5.2f} TB/s", flush=True)
        out[tag] = {"gateup_us": gu[0], "gateup_min_us": gu[1], "down_us": dn[0], "down_min_us": dn[1]}

    gu, dn = time_pair(-4)
    report("stream ksplit=auto (default)", gu, dn)
    gu, dn = time_pair(2)
    report("stream ksplit=2", gu, dn)
    if args.ab:
        # interleaved A/B of the kernel variants in ONE process (guide 5.4 rule 24): parity of each against the fp64 sample, then rounds
        for cfg in [int(c) for c in args.ab.split(",")]:
            _lib.tune("ps_cfg", cfg)
            hh, hl, y = run_ps(layers[0], -4)
            torch.cuda.synchronize()
            h = hh.float() + hl.float()
            ysum = y.sum(0)
            e_h = e_y = 0.0
            for pp in list(range(0, 2 * S, max(1, (2 * S) // 48))) + [2 * S - 1] + [int(v) - 1 for v in np.cumsum(rows) if v > 0]:
                slot = int(order[pp]); e = int(flat[slot]); t = slot // 2
                xr = x[t].double()
                gg = L["w1"][e].double() @ xr
                uu = L["w3"][e].double() @ xr
                href = gg / (1 + torch.exp(-gg)) * uu
                e_h = max(e_h, float(torch.nan_to_num((h[pp].double() - href).abs(), nan=1e30).max()))
                e_y = max(e_y, float(torch.nan_to_num((ysum[slot].double() - L["w2"][e].double() @ h[pp].double()).abs(), nan=1e30).max()))
            print(f"cfg={cfg}: max |err| gate/up {e_h:.3e}   down {e_y:.3e}   (K split {int(nslab.item())})", flush=True)
            assert args.nocheck or (e_h < 5e-4 and e_y < 5e-4), f"parity failure cfg={cfg}"
        for rnd in range(args.rounds):
            for cfg in [int(c) for c in args.ab.split(",")]:
                _lib.tune("ps_cfg", cfg)
                gu, dn = time_pair(-4)
                report(f"round {rnd} cfg={cfg}", gu, dn)
        _lib.tune("ps_cfg", -1)
    if args.abtune:
        key, vals = args.abtune.split("=")
        vals = [int(v) for v in vals.split(",")]
        for v in vals:
            _lib.tune(key, v)
            hh, hl, y = run_ps(layers[0], -4)
            torch.cuda.synchronize()
            h = hh.float() + hl.float()
            ysum = y.sum(0)
            e_h = e_y = 0.0
            for pp in list(range(0, 2 * S, max(1, (2 * S) // 48))) + [2 * S - 1] + [int(c) - 1 for c in np.cumsum(rows) if c > 0]:
                slot = int(order[pp]); e = int(flat[slot]); t = slot // 2
                xr = x[t].double()
                gg = L["w1"][e].double() @ xr
                uu = L["w3"][e].double() @ xr
                href = gg / (1 + torch.exp(-gg)) * uu
                e_h = max(e_h, float(torch.nan_to_num((h[pp].double() - href).abs(), nan=1e30).max()))
                e_y = max(e_y, float(torch.nan_to_num((ysum[slot].double() - L["w2"][e].double() @ h[pp].double()).abs(), nan=1e30).max()))
            print(f"{key}={v}: max |err| gate/up {e_h:.3e}   down {e_y:.3e}   (K split {int(nslab.item())})", flush=True)
            assert args.nocheck or (e_h < 5e-4 and e_y < 5e-4), f"parity failure {key}={v}"
        for rnd in range(args.rounds):
            for v in vals:
                _lib.tune(key, v)
                gu, dn = time_pair(-4)
                report(f"round {rnd} {key}={v}", gu, dn)
        _lib.tune(key, -1)
    if args.sweep:
        gu, dn = time_pair(1)
        report("stream ksplit=1", gu, dn)
        for cfg in (0, 1):
            _lib.tune("ps_cfg", cfg)
            gu, dn = time_pair(2)
            report(f"stream cfg={cfg}", gu, dn)
        _lib.tune("ps_cfg", -1)
        _lib.tune("ps_nt", 0)
        gu, dn = time_pair(2)
        report("stream, default-policy weight loads", gu, dn)
        _lib.tune("ps_nt", -1)

        def ggu(L):
            ops.gemm(x, L["w1"], w_up=L["w3"], a_rowidx=stok, group_off=goff, ngroups=E, w_group_stride=I * H, m=2 * S)

        h0, _ = run_general(layers[0])
        yy = torch.empty((2 * S, H), dtype=torch.float32, device=dev)

        def gdn(L):
            ops.gemm(h0, L["w2"], group_off=goff, ngroups=E, w_group_stride=H * I, c_rowidx=sslot, out=yy)

        report("general kernel (round 1)", timed(ggu, args.iters), timed(gdn, args.iters))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
