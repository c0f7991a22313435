"""ORACLE (test infrastructure, not product): fp32 numpy restatement of the Mixtral backbone the
reference reaches through HF transformers (pinned transformers==4.41.1, reference
requirements.txt:24; arithmetic not vendored in /root/reference — SURVEY F4).  Call sites
restated: vita/model/language_model/vita_mixtral.py:158-173 (model(...) then lm_head, no fp32
upcast), video_audio_demo.py:257-270 (greedy generate).  Function-by-function sources are the
HF file transformers/models/mixtral/modeling_mixtral.py (container copy 5.15.0; 4.41.1 deltas
listed in SURVEY §8(c) are neutral in fp32):

  rmsnorm            MixtralRMSNorm.forward                 :143-148
  rope_cos_sin       MixtralRotaryEmbedding                 :169-201
  apply_rope         rotate_half / apply_rotary_pos_emb     :203-241
  attention          eager_attention_forward + repeat_kv    :244-279
  moe                MixtralTopKRouter + MixtralExperts     :57-111
  decoder_layer      MixtralDecoderLayer.forward
  forward / greedy   MixtralModel.forward + argmax loop

Pinned against the installed HF implementation by tests/test_oracle_pin.py (CPU) and against
golden vectors generated from it (tests/golden/, oracle/make_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import numpy as np

F32 = np.float32


def rmsnorm(x, w, eps):
    x = x.astype(F32)
    var = np.mean(x * x, axis=-1, keepdims=True, dtype=F32)
    return (w * (x * (1.0 / np.sqrt(var + F32(eps))).astype(F32))).astype(F32)


def rope_cos_sin(positions, head_dim, theta):
    inv_freq = 1.0 / (np.float64(theta) ** (np.arange(0, head_dim, 2, dtype=np.float64) / head_dim))
    inv_freq = inv_freq.astype(F32)  # HF keeps inv_freq in fp32, then an fp32 outer product
    freqs = (np.asarray(positions, dtype=F32)[:, None] * inv_freq[None, :]).astype(F32)
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(F32), np.sin(emb).astype(F32)


def rotate_half(x):
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def apply_rope(x, cos, sin):
    """x [heads, S, d]; cos/sin [S, d]."""
    return (x * cos[None] + rotate_half(x) * sin[None]).astype(F32)


def softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp((x - m).astype(F32))
    return (e / np.sum(e, axis=axis, keepdims=True, dtype=F32)).astype(F32)


def attention(q, k, v, q_pos0):
    """q [nq, Sq, d]; k, v [nkv, Sk, d] (whole cache incl. current); causal: key j visible to query i
    iff j <= q_pos0 + i."""
    nq, Sq, d = q.shape
    nkv, Sk, _ = k.shape
    g = nq // nkv
    out = np.empty((Sq, nq * d), F32)
    scale = F32(d) ** F32(-0.5)
    mask = np.arange(Sk)[None, :] > (q_pos0 + np.arange(Sq))[:, None]
    for h in range(nq):
        s = (q[h] @ k[h // g].T).astype(F32) * scale
        s = np.where(mask, F32(-np.inf), s)
        p = softmax(s)
        out[:, h * d:(h + 1) * d] = p @ v[h // g]
    return out


def router(x, gate_w, top_k=2):
    logits = (x @ gate_w.T).astype(F32)
    probs = softmax(logits)
    idx = np.argsort(-probs, axis=-1, kind="stable")[:, :top_k]
    val = np.take_along_axis(probs, idx, axis=-1)
    val = (val / np.sum(val, axis=-1, keepdims=True, dtype=F32)).astype(F32)
    return idx, val


def silu(x):
    return (x / (1.0 + np.exp(-x))).astype(F32)


def moe(x, lw, top_k=2, force=None):
    """x [S, H].  lw: gate [E,H], w1/w3 [E,I,H], w2 [E,H,I].
    force (test infrastructure, default off): {row: (e_a, e_b)} — take THESE experts for a row instead of the router's top-k (their
    weights are the row's own softmax probabilities, renormalised): used where the router's own margin is a tie at fp32 noise and a
    checked implementation legitimately took the other expert, to follow that branch of the (ill-conditioned) reference."""
    idx, val = router(x, lw["gate"], top_k)
    if force:
        probs = softmax((x @ lw["gate"].T).astype(F32))
        idx, val = idx.copy(), val.copy()
        for row, ids in force.items():
            ids = np.asarray(ids, dtype=idx.dtype)
            pv = probs[row, ids]
            idx[row] = ids
            val[row] = (pv / np.sum(pv, dtype=F32)).astype(F32)
    out = np.zeros_like(x, dtype=F32)
    for e in range(lw["gate"].shape[0]):
        tok, slot = np.nonzero(idx == e)
        if tok.size == 0:
            continue
        cur = x[tok]
        h = silu(cur @ lw["w1"][e].T) * (cur @ lw["w3"][e].T)
        y = (h.astype(F32) @ lw["w2"][e].T).astype(F32) * val[tok, slot][:, None]
        np.add.at(out, tok, y.astype(F32))
    return out, idx, val


class MixtralOracle:
    """Weights from a reference-named state dict (float32 arrays)."""

    def __init__(self, sd, tcfg):
        self.t = tcfg
        t = tcfg
        g = lambda k: np.asarray(sd[k], F32)
        self.embed = g("model.embed_tokens.weight")
        self.norm = g("model.norm.weight")
        self.lm_head = g("lm_head.weight")
        self.layers = []
        for l in range(t.num_hidden_layers):
            p = f"model.layers.{l}."
            ex = lambda nm: np.stack([g(p + f"block_sparse_moe.experts.{e}.{nm}.weight")
                                      for e in range(t.num_local_experts)])
            self.layers.append(dict(
                ln1=g(p + "input_layernorm.weight"), q=g(p + "self_attn.q_proj.weight"),
                k=g(p + "self_attn.k_proj.weight"), v=g(p + "self_attn.v_proj.weight"),
                o=g(p + "self_attn.o_proj.weight"), ln2=g(p + "post_attention_layernorm.weight"),
                gate=g(p + "block_sparse_moe.gate.weight"), w1=ex("w1"), w2=ex("w2"), w3=ex("w3")))
        self.reset()

    def reset(self):
        self.kc = [None] * self.t.num_hidden_layers
        self.vc = [None] * self.t.num_hidden_layers
        self.pos = 0
        self.last_route = []

    def forward(self, x, want_hidden=False):
        """x [S, H] embeddings at positions [pos, pos+S).  Returns logits of all S rows (fp32) and
        optionally the residual stream after each layer."""
        t = self.t
        S = x.shape[0]
        d, nq, nkv = t.head_dim, t.num_attention_heads, t.num_key_value_heads
        cos, sin = rope_cos_sin(np.arange(self.pos, self.pos + S), d, t.rope_theta)
        hid = []
        self.last_route = []
        x = x.astype(F32)
        for l, L in enumerate(self.layers):
            xn = rmsnorm(x, L["ln1"], t.rms_norm_eps)
            q = (xn @ L["q"].T).astype(F32).reshape(S, nq, d).transpose(1, 0, 2)
            k = (xn @ L["k"].T).astype(F32).reshape(S, nkv, d).transpose(1, 0, 2)
            v = (xn @ L["v"].T).astype(F32).reshape(S, nkv, d).transpose(1, 0, 2)
            q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
            self.kc[l] = k if self.kc[l] is None else np.concatenate([self.kc[l], k], axis=1)
            self.vc[l] = v if self.vc[l] is None else np.concatenate([self.vc[l], v], axis=1)
            a = attention(q, self.kc[l], self.vc[l], self.pos)
            x = (x + a @ L["o"].T).astype(F32)
            xn = rmsnorm(x, L["ln2"], t.rms_norm_eps)
            y, idx, val = moe(xn, L, t.num_experts_per_tok)
            self.last_route.append((idx, val))
            x = (x + y).astype(F32)
            if want_hidden:
                hid.append(x.copy())
        self.pos += S
        logits = (rmsnorm(x, self.norm, t.rms_norm_eps) @ self.lm_head.T).astype(F32)
        return logits, (np.stack(hid) if want_hidden else None)

    def greedy(self, embeds, n_new, eos=None):
        """Hand-rolled greedy loop (argmax of the last row; the reference's generate() wrapper is
        broken under transformers 5.x — SURVEY H4).  Returns (ids, per-step last-row logits)."""
        self.reset()
        logits, _ = self.forward(embeds)
        ids, lg = [], []
        for _ in range(n_new):
            last = logits[-1]
            tok = int(np.argmax(last))
            ids.append(tok)
            lg.append(last.copy())
            if eos is not None and tok == eos:
                break
            logits, _ = self.forward(self.embed[tok][None, :])
        return ids, np.stack(lg)
