#!/usr/bin/env python
"""Bisect helper: error of the real-width one-layer prefill (tests/test_mixtral_gpu.py::_run, S = 20, seed 6) against the oracle
under attention variants."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mixtral as om
from vita_amd import _lib
from vita_amd.checkpoint import pack_mixtral, synth_state_dict
from vita_amd.config import TextConfig, VitaConfig
from vita_amd.engine import MixtralEngine

dev = torch.device("cuda:0")
cfg = VitaConfig.tiny()
cfg.text = TextConfig(num_hidden_layers=1, vocab_size=4096)
for seed, S in ((6, 20), (5, 48), (7, 33)):
    sd = synth_state_dict(cfg, seed=seed, parts=("text",))
    rng = np.random.default_rng(seed + 100)
    ids_in = rng.integers(3, cfg.text.vocab_size, size=S)
    emb = sd["model.embed_tokens.weight"][ids_in]
    orc = om.MixtralOracle(sd, cfg.text)
    _, ref_hid = orc.forward(emb, want_hidden=True)
    packed = pack_mixtral(sd, cfg, dev)
    for v in ({"attn_impl": 0}, {"attn_impl": 2}):
        for k, val in v.items():
            _lib.tune(k, val)
        eng = MixtralEngine(cfg, packed, dev, max_ctx=S + 16, max_prefill=S, max_new=8)
        _, hid = eng.prefill(torch.from_numpy(emb).to(dev), want_hidden=True)
        torch.cuda.synchronize()
        d = np.abs(hid[0].cpu().numpy().astype(np.float64) - ref_hid[0])
        viol = d - (2e-4 + 1e-4 * np.abs(ref_hid[0]))
        print(f"seed {seed} S {S} {v}: max diff {d.max():.3e} mean {d.mean():.3e} worst margin {viol.max():.3e} rows over {sorted(set(np.argwhere(viol > 0)[:, 0].tolist()))}", flush=True)
        eng.close()
    _lib.tune("attn_impl", 0)
