#!/usr/bin/env python
"""The reference's OWN modules timed on CPU (SURVEY 8(d) "Reference CPU timing"): run in the BUILD container, where
/root/reference exists (the GPU box has no copy; bench.py's cpu_baseline times the numpy port there).
  encoders  vita.model.multimodal_encoder InternVisionModel (24 layers) + mlp2x_gelu projector on one 448x448 tile,
            whale audioEncoder + adapter on the 10 s clip — fp32, torch CPU
  backbone  the installed HF MixtralForCausalLM (the class VITAMixtralForCausalLM subclasses) with `--layers` real-geometry
            layers: prefill over the S=552 prompt, then greedy steps; extrapolated to 32 layers
Usage: python profiles/ref_cpu_timing.py [--layers 2] [--threads 8] > profiles/r02_reference_cpu_timing.json"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hashw, hf_mixtral, ref_harness as rh  # noqa: E402
from vita_amd.checkpoint import synth_state_dict  # noqa: E402
from vita_amd.config import VitaConfig  # noqa: E402
from vita_amd.host.synthetic import make_request  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    cfg = VitaConfig()
    req = make_request(cfg)
    rh.install()
    out = {"where": "build container", "cores": args.threads, "dtype": "fp32 torch CPU", "kind": "reference"}
    sd = synth_state_dict(cfg, seed=1, rich=False, parts=("vision", "audio"))
    tower, pj, aud = rh.build_internvit(cfg, sd), rh.build_projector(cfg, sd), rh.build_whale(cfg, sd)
    pix = torch.from_numpy(req["pixel_values"])
    feats = torch.from_numpy(req["fbank"])[None]
    with torch.no_grad():
        tower(pix)                                            # warm-up
        t0 = time.perf_counter(); v = tower(pix); pj(v); out["vit_projector_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        aud(feats, torch.tensor([feats.shape[1]]))
        t0 = time.perf_counter(); aud(feats, torch.tensor([feats.shape[1]])); out["audio_encoder_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    del tower, pj, aud, sd
    t = copy.deepcopy(cfg.text)
    L_full, t.num_hidden_layers = t.num_hidden_layers, args.layers
    H = t.hidden_size
    names = {"model.embed_tokens.weight": (t.vocab_size, H), "lm_head.weight": (t.vocab_size, H)}
    for l in range(args.layers):
        p = f"model.layers.{l}."
        names[p + "self_attn.q_proj.weight"] = (t.num_attention_heads * t.head_dim, H)
        names[p + "self_attn.k_proj.weight"] = (t.num_key_value_heads * t.head_dim, H)
        names[p + "self_attn.v_proj.weight"] = (t.num_key_value_heads * t.head_dim, H)
        names[p + "self_attn.o_proj.weight"] = (H, t.num_attention_heads * t.head_dim)
        names[p + "block_sparse_moe.gate.weight"] = (t.num_local_experts, H)
        for e in range(t.num_local_experts):
            q = p + f"block_sparse_moe.experts.{e}."
            names[q + "w1.weight"], names[q + "w3.weight"], names[q + "w2.weight"] = (t.intermediate_size, H), (t.intermediate_size, H), (H, t.intermediate_size)
    sdt = {k: hashw.fill(shape, hashw.tensor_seed(k, 0)) for k, shape in names.items()}
    sdt["model.norm.weight"] = np.ones(H, np.float32)
    for l in range(args.layers):
        sdt[f"model.layers.{l}.input_layernorm.weight"] = np.ones(H, np.float32)
        sdt[f"model.layers.{l}.post_attention_layernorm.weight"] = np.ones(H, np.float32)
    m = hf_mixtral.build(t, sdt)
    del sdt
    S = 552
    x = torch.from_numpy(hashw.fill((S, H), 12345))[None]
    with torch.no_grad():
        t0 = time.perf_counter()
        o = m(inputs_embeds=x, use_cache=True)
        t_pre = time.perf_counter() - t0
        past, logits = o.past_key_values, o.logits[0, -1]
        ts = []
        for _ in range(args.steps):
            tok = int(torch.argmax(logits))
            t0 = time.perf_counter()
            o = m(input_ids=torch.tensor([[tok]]), past_key_values=past, use_cache=True)
            ts.append(time.perf_counter() - t0)
            past, logits = o.past_key_values, o.logits[0, -1]
        t0 = time.perf_counter()
        m.lm_head(m.model.norm(torch.zeros(1, 1, H)))
        t_head = time.perf_counter() - t0
    step = float(np.median(ts[1:])) if len(ts) > 1 else ts[0]
    per_layer = max(step - t_head, 1e-9) / args.layers
    out.update({"prefill_ms_extrapolated": round(((t_pre - t_head) / args.layers * L_full + t_head) * 1e3, 1),
                "decode_tokens_per_s_extrapolated": round(1.0 / (per_layer * L_full + t_head), 4),
                "ms_per_layer_token": round(per_layer * 1e3, 3), "ms_lm_head": round(t_head * 1e3, 3),
                "sample": f"HF MixtralForCausalLM (installed transformers), {args.layers} of {L_full} real-geometry layers, prompt S={S}, "
                          f"{args.steps} greedy steps, extrapolated x{L_full}/{args.layers}"})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
