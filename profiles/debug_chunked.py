#!/usr/bin/env python
"""Bisect helper: chunked prefill (83 + 67 rows, pos0 = 83) against one-shot on a real-width engine of a few layers, under
kernel-variant knobs.  python profiles/debug_chunked.py [--layers 2]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vita_amd import _lib
from vita_amd.checkpoint import synth_mixtral_device
from vita_amd.config import VitaConfig
from vita_amd.engine import MixtralEngine

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = VitaConfig()
cfg.text.num_hidden_layers = args.layers
packed = synth_mixtral_device(cfg, dev, seed=0)
rng = np.random.default_rng(1)
ids = rng.integers(3, cfg.text.vocab_size, size=150).tolist()
emb = lambda i: packed["embed"][torch.as_tensor(i, device=dev)].float()
DEF = {"attn_impl": 0, "prefill_fuse_rows": 1, "attn_fa": 1, "ps_cfg": -1}
for v in ({}, {"attn_impl": 2}, {"prefill_fuse_rows": 0}, {"attn_fa": 0}, {"ps_cfg": 1}):
    for k, val in {**DEF, **v}.items():
        _lib.tune(k, val)
    eng = MixtralEngine(cfg, packed, dev, max_ctx=512, max_prefill=256, max_new=8)
    one, h1 = eng.prefill(emb(ids), want_hidden=True)
    one, h1 = one.clone(), h1.clone()
    eng.prefill(emb(ids[:83]))
    two, h2 = eng.prefill(emb(ids[83:]), pos0=83, want_hidden=True)
    torch.cuda.synchronize()
    errs = [float((h1[l][83:] - h2[l]).abs().max()) for l in range(args.layers)]
    print(v or "default", "logits diff %.2e" % float((one - two).abs().max()), "hidden diff per layer", ["%.1e" % e for e in errs], flush=True)
    eng.close()
