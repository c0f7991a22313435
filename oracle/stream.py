"""ORACLE (test infrastructure): the Mixtral backbone at the RELEASED geometry, one layer at a time.

The full fp32 model is 187 GB; a host holds one layer (5.7 GB).  Weights come from the counter-based generator
(oracle/hashw.py — the device fills its bf16 copy with the same integers, vita_amd/checkpoint.py
synth_mixtral_device), keyed by the reference's parameter names, so nothing has to be stored or copied.  The
arithmetic is oracle/mixtral.py's (the pinned restatement of HF modeling_mixtral.py as the reference drives it,
vita/model/language_model/vita_mixtral.py:158-173): one causal forward over prompt + teacher-forced tokens gives, for
every generated position, the logits a token-by-token greedy loop would have produced (same prefix => same logits),
which is what tests/test_realgeom_gpu.py compares with the device's prefill + decode steps.

Golden list of SURVEY 8(c): hidden states after chosen layers, router top-2 ids of every layer, logits rows, ids.
"""
import time

import numpy as np

from . import hashw
from . import mixtral as om

F32 = np.float32


class LayerBuffers:
    """Reusable host buffers for one decoder layer's weights (avoids re-faulting 5.7 GB per layer)."""

    def __init__(self, t):
        H, I, E, hd = t.hidden_size, t.intermediate_size, t.num_local_experts, t.head_dim
        nq, nkv = t.num_attention_heads, t.num_key_value_heads
        self.q = np.empty((nq * hd, H), F32)
        self.k = np.empty((nkv * hd, H), F32)
        self.v = np.empty((nkv * hd, H), F32)
        self.o = np.empty((H, nq * hd), F32)
        self.gate = np.empty((E, H), F32)
        self.w1 = np.empty((E, I, H), F32)
        self.w3 = np.empty((E, I, H), F32)
        self.w2 = np.empty((E, H, I), F32)

    def load(self, t, l, seed):
        p = f"model.layers.{l}."
        s = lambda name: hashw.tensor_seed(p + name, seed)
        hashw.fill(self.q.shape, s("self_attn.q_proj.weight"), out=self.q)
        hashw.fill(self.k.shape, s("self_attn.k_proj.weight"), out=self.k)
        hashw.fill(self.v.shape, s("self_attn.v_proj.weight"), out=self.v)
        hashw.fill(self.o.shape, s("self_attn.o_proj.weight"), out=self.o)
        hashw.fill(self.gate.shape, s("block_sparse_moe.gate.weight"), out=self.gate)
        for e in range(t.num_local_experts):
            q = f"block_sparse_moe.experts.{e}."
            hashw.fill(self.w1[e].shape, s(q + "w1.weight"), out=self.w1[e])
            hashw.fill(self.w3[e].shape, s(q + "w3.weight"), out=self.w3[e])
            hashw.fill(self.w2[e].shape, s(q + "w2.weight"), out=self.w2[e])
        ones = np.ones(t.hidden_size, F32)     # norm weights 1
        return dict(ln1=ones, ln2=ones, q=self.q, k=self.k, v=self.v, o=self.o, gate=self.gate, w1=self.w1, w3=self.w3,
                    w2=self.w2)


def embed_rows(t, ids, seed):
    """rows of model.embed_tokens.weight for the given ids (the table itself is never materialised)."""
    H = t.hidden_size
    s = hashw.tensor_seed("model.embed_tokens.weight", seed)
    out = np.empty((len(ids), H), F32)
    for i, tok in enumerate(ids):
        hashw.fill((H,), s, out=out[i], idx0=int(tok) * H)
    return out


def forward(t, seed, embeds, n_layers=None, capture=(), logits_from=0, verbose=False, margins=False, route_override=None,
            keep_inputs=False, resume=None):
    """One causal forward over embeds [S, H] (positions 0..S-1) through layers 0..n_layers-1 + final norm + LM head.
    Returns dict(logits [S - logits_from, V], hidden {layer: [S, H]}, route [n_layers, S, 2] int32); margins=True adds
    margin [n_layers, S]: the router-logit distance between the 2nd and the 3rd expert (how far each top-2 decision is
    from flipping).  route_override {layer: {row: (e_a, e_b)}}: follow a given expert pair at a TIED decision (om.moe force);
    keep_inputs: also return inputs {layer: x at the layer's entry}; resume (layer, x): start at that layer from that state (the
    rows of `route` / `margin` of the skipped layers are left unset)."""
    n_layers = t.num_hidden_layers if n_layers is None else n_layers
    S = embeds.shape[0]
    d, nq, nkv = t.head_dim, t.num_attention_heads, t.num_key_value_heads
    cos, sin = om.rope_cos_sin(np.arange(S), d, t.rope_theta)
    bufs = LayerBuffers(t)
    x = embeds.astype(F32)
    hidden, route = {}, np.empty((n_layers, S, 2), np.int32)
    margin = np.empty((n_layers, S), F32) if margins else None
    inputs = {}
    l_start = 0
    if resume is not None:
        l_start, x = int(resume[0]), np.asarray(resume[1], F32).copy()
    for l in range(l_start, n_layers):
        t0 = time.time()
        if keep_inputs:
            inputs[l] = x.copy()
        L = bufs.load(t, l, seed)
        t1 = time.time()
        xn = om.rmsnorm(x, L["ln1"], t.rms_norm_eps)
        q = (xn @ L["q"].T).astype(F32).reshape(S, nq, d).transpose(1, 0, 2)
        k = (xn @ L["k"].T).astype(F32).reshape(S, nkv, d).transpose(1, 0, 2)
        v = (xn @ L["v"].T).astype(F32).reshape(S, nkv, d).transpose(1, 0, 2)
        q, k = om.apply_rope(q, cos, sin), om.apply_rope(k, cos, sin)
        a = om.attention(q, k, v, 0)
        x = (x + a @ L["o"].T).astype(F32)
        xn = om.rmsnorm(x, L["ln2"], t.rms_norm_eps)
        y, idx, _ = om.moe(xn, L, t.num_experts_per_tok, force=(route_override or {}).get(l))
        route[l] = idx
        if margins:
            rl = np.sort((xn @ L["gate"].T).astype(F32), axis=-1)
            margin[l] = rl[:, -2] - rl[:, -3]
        x = (x + y).astype(F32)
        if l in capture:
            hidden[l] = x.copy()
        if verbose:
            print(f"[oracle] layer {l}: weights {t1 - t0:.1f}s compute {time.time() - t1:.1f}s", flush=True)
    norm = np.ones(t.hidden_size, F32)
    lm = hashw.fill((t.vocab_size, t.hidden_size), hashw.tensor_seed("lm_head.weight", seed))
    logits = (om.rmsnorm(x[logits_from:], norm, t.rms_norm_eps) @ lm.T).astype(F32)
    out = dict(logits=logits, hidden=hidden, route=route)
    if margins:
        out["margin"] = margin
    if keep_inputs:
        out["inputs"] = inputs
    return out


def forward_many(t, seed, embeds_list, n_layers=None, logits_from=None):
    """forward() for SEVERAL independent sequences with every layer's weights generated once (a layer is 5.7 GB of fp32: the
    generator, not the arithmetic, is what a short sequence pays for).  embeds_list: [S_i, H] arrays; logits_from: per-sequence
    first row whose logits are wanted (default 0).  Returns a list of dict(logits [S_i - from_i, V], route [n_layers, S_i, 2])."""
    n_layers = t.num_hidden_layers if n_layers is None else n_layers
    n = len(embeds_list)
    logits_from = [0] * n if logits_from is None else list(logits_from)
    d, nq, nkv = t.head_dim, t.num_attention_heads, t.num_key_value_heads
    bufs = LayerBuffers(t)
    xs = [e.astype(F32) for e in embeds_list]
    routes = [np.empty((n_layers, x.shape[0], 2), np.int32) for x in xs]
    rope = [om.rope_cos_sin(np.arange(x.shape[0]), d, t.rope_theta) for x in xs]
    for l in range(n_layers):
        L = bufs.load(t, l, seed)
        for i, x in enumerate(xs):
            S = x.shape[0]
            cos, sin = rope[i]
            xn = om.rmsnorm(x, L["ln1"], t.rms_norm_eps)
            q = (xn @ L["q"].T).astype(F32).reshape(S, nq, d).transpose(1, 0, 2)
            k = (xn @ L["k"].T).astype(F32).reshape(S, nkv, d).transpose(1, 0, 2)
            v = (xn @ L["v"].T).astype(F32).reshape(S, nkv, d).transpose(1, 0, 2)
            q, k = om.apply_rope(q, cos, sin), om.apply_rope(k, cos, sin)
            x = (x + om.attention(q, k, v, 0) @ L["o"].T).astype(F32)
            y, idx, _ = om.moe(om.rmsnorm(x, L["ln2"], t.rms_norm_eps), L, t.num_experts_per_tok)
            routes[i][l] = idx
            xs[i] = (x + y).astype(F32)
    norm = np.ones(t.hidden_size, F32)
    lm = hashw.fill((t.vocab_size, t.hidden_size), hashw.tensor_seed("lm_head.weight", seed))
    return [dict(logits=(om.rmsnorm(x[f:], norm, t.rms_norm_eps) @ lm.T).astype(F32), route=r) for x, f, r in zip(xs, logits_from, routes)]
