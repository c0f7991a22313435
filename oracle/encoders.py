"""ORACLE (test infrastructure, not product): fp32/fp64 numpy restatement of the two encoder
towers, the projector and the multimodal splice of the reference.  Each function cites the
reference lines it follows (paths relative to /root/reference).  Pinned against the reference's
own modules (imported with shims in THIS container) by oracle/make_golden.py, which also writes
the golden vectors in tests/golden/ that the GPU box replays.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
import math

import numpy as np
from scipy.special import erf

F = np.float64  # the oracle accumulates in fp64; inputs/weights are the same fp32/bf16 values


class precision:
    """`with precision(np.float32): ...` runs the restatements below in another working dtype.  The CHECKER is fp64 (the default;
    tests and smoke never change it); bench.py's cpu_baseline times the fp32 form — single-precision BLAS on the same arithmetic is
    what a CPU implementation of the reference's fp32 modules runs, the fp64 form is 2-3x slower for no reason a baseline should carry."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global F
        self.prev, F = F, self.dtype
        return self

    def __exit__(self, *exc):
        global F
        F = self.prev
        return False


def layernorm(x, w, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def gelu(x):  # nn.GELU() default = erf form
    return 0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))


def softmax(x):
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(-1, keepdims=True)


# ---------------------------------------------------------------------------------------------
# InternViT-300M tower + pixel shuffle
# ---------------------------------------------------------------------------------------------
def internvit_embeddings(sd, v, pix):
    """InternVisionEmbeddings.forward — vita/model/multimodal_encoder/internvit/modeling_intern_vit.py:107-122.
    Conv2d(3,C,14,14) == per-patch matmul; pos-embed bicubic resize is the identity when the patch
    grid equals the trained grid (:92-105), the only case restated."""
    P = "model.vision_tower.vision_tower.embeddings."
    n, _, Hh, Ww = pix.shape
    ps = v.patch_size
    gh, gw = Hh // ps, Ww // ps
    assert gh == gw == v.grid, "pos-embed interpolation not restated (identity case only)"
    w = sd[P + "patch_embedding.weight"].astype(F, copy=False).reshape(v.hidden_size, -1)
    patches = pix.astype(F, copy=False).reshape(n, 3, gh, ps, gw, ps).transpose(0, 2, 4, 1, 3, 5).reshape(n, gh * gw, -1)
    pe = patches @ w.T + sd[P + "patch_embedding.bias"].astype(F, copy=False)
    cls = np.broadcast_to(sd[P + "class_embedding"].astype(F, copy=False), (n, 1, v.hidden_size))
    return np.concatenate([cls, pe], 1) + sd[P + "position_embedding"].astype(F, copy=False)


def internvit_layer(sd, v, l, x):
    """InternVisionEncoderLayer.forward (:237-253) with _naive_attn (:158-177) and InternMLP (:213-217)."""
    p = f"model.vision_tower.vision_tower.encoder.layers.{l}."
    g = lambda k: sd[p + k].astype(F, copy=False)
    n, N, C = x.shape
    nh = v.num_attention_heads
    d = C // nh
    h = layernorm(x, g("norm1.weight"), g("norm1.bias"), v.layer_norm_eps)
    qkv = (h @ g("attn.qkv.weight").T + g("attn.qkv.bias")).reshape(n, N, 3, nh, d).transpose(2, 0, 3, 1, 4)
    q, k, vv = qkv[0], qkv[1], qkv[2]
    att = softmax((q * d ** -0.5) @ k.transpose(0, 1, 3, 2))
    a = (att @ vv).transpose(0, 2, 1, 3).reshape(n, N, C)
    a = a @ g("attn.proj.weight").T + g("attn.proj.bias")
    x = x + a * g("ls1")
    h = layernorm(x, g("norm2.weight"), g("norm2.bias"), v.layer_norm_eps)
    m = gelu(h @ g("mlp.fc1.weight").T + g("mlp.fc1.bias")) @ g("mlp.fc2.weight").T + g("mlp.fc2.bias")
    return x + m * g("ls2")


def pixel_shuffle(x, scale=0.5):
    """InternViTVisionTower.pixel_shuffle — internvit_encoder.py:42-53 (same view/permute chain)."""
    n, w, h, c = x.shape
    x = x.reshape(n, w, int(h * scale), int(c / scale)).transpose(0, 2, 1, 3)
    x = x.reshape(n, int(h * scale), int(w * scale), int(c / (scale * scale))).transpose(0, 2, 1, 3)
    return x


def internvit_tower(sd, v, pix, want_layers=False):
    """InternViTVisionTower.forward — internvit_encoder.py:55-79: last hidden state, drop CLS (:35-40),
    x0.5, pixel shuffle, flatten to [n, 256, 4096]."""
    x = internvit_embeddings(sd, v, pix)
    layers = []
    for l in range(v.num_hidden_layers):
        x = internvit_layer(sd, v, l, x)
        if want_layers:
            layers.append(x.copy())
    f = x[:, 1:]
    g = int(round(math.sqrt(f.shape[1])))
    assert g * g == f.shape[1]
    f = pixel_shuffle(f.reshape(f.shape[0], g, g, -1) * 0.5)
    out = f.reshape(f.shape[0], -1, f.shape[-1])
    return (out, layers) if want_layers else out


def projector(sd, feats):
    """mlp2x_gelu — vita/model/multimodal_projector/builder.py:154-168: Linear, GELU, Linear."""
    g = lambda k: sd["model.mm_projector." + k].astype(F, copy=False)
    return gelu(feats @ g("0.weight").T + g("0.bias")) @ g("2.weight").T + g("2.bias")


# ---------------------------------------------------------------------------------------------
# Whale audio encoder + CNNSubsampling adapter
# ---------------------------------------------------------------------------------------------
def sinusoid_pe(length, d):
    """PositionalEncoding table — whale/module/layer/attention.py:26-37, fp32 op for op."""
    pos = np.arange(0, length, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / d))).astype(np.float32)
    ang = (pos * div).astype(np.float32)
    pe = np.zeros((length, d), np.float32)
    pe[:, 0::2] = np.sin(ang)
    pe[:, 1::2] = np.cos(ang)
    return pe


def chunk_mask(size, chunk, left):
    """subsequent_chunk_mask — whale/utils.py:88-103."""
    ret = np.zeros((size, size), bool)
    for i in range(size):
        start = 0 if left < 0 else max((i // chunk - left) * chunk, 0)
        ret[i, start:min((i // chunk + 1) * chunk, size)] = True
    return ret


def conv2d_valid_s2(x, w, b):
    """x [Cin, T, Fq]; w [Cout, Cin, 3, 3]; stride 2, no padding."""
    Cin, T, Fq = x.shape
    To, Fo = (T - 3) // 2 + 1, (Fq - 3) // 2 + 1
    cols = np.empty((To, Fo, Cin, 3, 3), F)
    for kh in range(3):
        for kw in range(3):
            cols[:, :, :, kh, kw] = x[:, kh:kh + 2 * To:2, kw:kw + 2 * Fo:2].transpose(1, 2, 0)
    out = cols.reshape(To * Fo, -1) @ w.reshape(w.shape[0], -1).T + b
    return out.reshape(To, Fo, -1).transpose(2, 0, 1)


def whale_encoder(sd, a, feats, length=None, chunk=0, left=-1, want_layers=False, dbg=None):
    """audioEncoder.forward for ONE utterance — whale/init_model.py:114-139.
    feats [T, 80] fp32.  Full attention unless chunk > 0 (hazard H1: the reference's random
    dynamic-chunk mask is pinned off; an explicit (chunk, left) mask is supported)."""
    A = "model.audio_encoder."
    g = lambda k: sd[A + k].astype(F, copy=False)
    T = feats.shape[0]
    length = T if length is None else int(length)
    # whaleEncoder.forward — module/encoder/encoder.py:140-147: pad mask, GlobalCMVN (cmvn.py:29-32)
    x = (feats.astype(F, copy=False) - g("encoder.global_cmvn.mean")) * g("encoder.global_cmvn.istd")
    mask = np.arange(T) < length
    # Conv2dSubsampling4.forward — module/component/subsampling.py:38-43
    c = "encoder.enc.0.core."
    y = np.maximum(conv2d_valid_s2(x[None], g(c + "conv.0.weight"), g(c + "conv.0.bias")), 0)
    if dbg is not None:
        dbg["conv1"] = y.transpose(1, 2, 0).copy()          # [T1, F1, C]
    y = np.maximum(conv2d_valid_s2(y, g(c + "conv.2.weight"), g(c + "conv.2.bias")), 0)
    C, T2, F2 = y.shape
    if dbg is not None:
        dbg["conv2"] = y.transpose(1, 2, 0).copy()          # [T2, F2, C]
    y = y.transpose(1, 0, 2).reshape(T2, C * F2) @ g(c + "out.0.weight").T + g(c + "out.0.bias")
    if dbg is not None:
        dbg["sub_out"] = y.copy()
    mask = mask[2::2][2::2]
    # Transformer.forward — module/component/transformer.py:374-394
    e = "encoder.enc.1."
    amask = np.broadcast_to(mask[None, :], (T2, T2)).copy()
    if chunk > 0:
        amask &= chunk_mask(T2, chunk, left)
    y = np.maximum(layernorm(y @ g(e + "embed.0.weight").T + g(e + "embed.0.bias"), g(e + "embed.1.weight"),
                             g(e + "embed.1.bias"), 1e-5), 0)
    y = y * math.sqrt(C)                                    # RelPositionalEncoding.forward — attention.py:100-111
    if dbg is not None:
        dbg["embed"] = y.copy()
    pos = sinusoid_pe(T2, C).astype(F, copy=False)
    nh = a.num_attention_heads
    dk = C // nh
    layers = []
    for l in range(a.num_hidden_layers):
        p = e + f"encoders.{l}."
        # TransformerLayer.forward (pre-norm) — transformer.py:100-125
        h = layernorm(y, g(p + "norm1.weight"), g(p + "norm1.bias"), a.layer_norm_eps)
        # MultiHeadedAttention.forward — attention.py:358-419 (rel-pos, NO rel_shift :395-397)
        q = (h @ g(p + "self_attn.linear_q.weight").T + g(p + "self_attn.linear_q.bias")).reshape(T2, nh, dk)
        k = (h @ g(p + "self_attn.linear_k.weight").T + g(p + "self_attn.linear_k.bias")).reshape(T2, nh, dk)
        v = (h @ g(p + "self_attn.linear_v.weight").T + g(p + "self_attn.linear_v.bias")).reshape(T2, nh, dk)
        pp = (pos @ g(p + "self_attn.linear_pos.weight").T).reshape(T2, nh, dk)
        qu = (q + g(p + "self_attn.pos_bias_u")).transpose(1, 0, 2)
        qv = (q + g(p + "self_attn.pos_bias_v")).transpose(1, 0, 2)
        sc = (qu @ k.transpose(1, 2, 0) + qv @ pp.transpose(1, 2, 0)) / math.sqrt(dk)
        sc = np.where(amask[None], sc, float(np.finfo(np.float16).min))      # masked_fill(min_value) :404-406
        att = np.where(amask[None], softmax(sc), 0.0)                          # .masked_fill(mask, 0.0) :407-409
        o = (att @ v.transpose(1, 0, 2)).transpose(1, 0, 2).reshape(T2, C)
        y = y + (o @ g(p + "self_attn.linear_out.weight").T + g(p + "self_attn.linear_out.bias"))
        h = layernorm(y, g(p + "norm2.weight"), g(p + "norm2.bias"), a.layer_norm_eps)
        # PositionwiseFeedForward — attention.py:145-147
        y = y + (np.maximum(h @ g(p + "feed_forward.w_1.weight").T + g(p + "feed_forward.w_1.bias"), 0)
                 @ g(p + "feed_forward.w_2.weight").T + g(p + "feed_forward.w_2.bias"))
        if want_layers:
            layers.append(y.copy())
    y = layernorm(y, g(e + "after_norm.weight"), g(e + "after_norm.bias"), a.layer_norm_eps)
    # CNNSubsampling.forward (single-conv branch) — adapter.py:107-136
    ad = "adpter."
    y = np.where(mask[:, None], y, 0.0)
    kz = a.adapter_kernel
    yp = np.concatenate([y, np.zeros((kz - 1, C), F)], 0)
    T3 = (T2 + kz - 1 - kz) // 2 + 1
    w = g(ad + "conv1d2.weight")  # [2C, C, k]
    z = np.stack([np.einsum("kc,ock->o", yp[2 * t:2 * t + kz], w) for t in range(T3)]) + g(ad + "conv1d2.bias")
    z = gelu(layernorm(z, g(ad + "bn2.weight"), g(ad + "bn2.bias"), a.adapter_norm_eps))
    z = z @ g(ad + "project.weight").T + g(ad + "project.bias")
    out_mask = mask[0::2]
    return (z, out_mask, layers) if want_layers else (z, out_mask)


# ---------------------------------------------------------------------------------------------
# multimodal splice (batch 1)
# ---------------------------------------------------------------------------------------------
IMAGE_TOKEN_INDEX, AUDIO_TOKEN_INDEX = -200, -500


def splice(input_ids, embed_table, image_feats, audio_feats, max_len=None):
    """prepare_inputs_labels_for_multimodal for one sequence — vita/model/vita_arch.py:237-329.
    input_ids: 1-D ints with -200 per image tile / -500 per audio clip; image_feats [n_tiles, 256, H];
    audio_feats [n_clips, T'', H].  Returns inputs_embeds [S, H]."""
    ids = np.asarray(input_ids)
    n_img, n_aud = int((ids == IMAGE_TOKEN_INDEX).sum()), int((ids == AUDIO_TOKEN_INDEX).sum())
    # count asserts — vita_arch.py:227-236
    assert n_img + (0 if n_img else 1) == image_feats.shape[0]
    assert n_aud + (0 if n_aud else 1) == audio_feats.shape[0]
    parts, ii, ai = [], 0, 0
    for t in ids:
        if t == IMAGE_TOKEN_INDEX:
            parts.append(image_feats[ii]); ii += 1
        elif t == AUDIO_TOKEN_INDEX:
            parts.append(audio_feats[ai]); ai += 1
        else:
            parts.append(embed_table[int(t)][None])
    out = np.concatenate(parts, 0) if parts else np.zeros((0, embed_table.shape[1]))
    if max_len is not None:
        out = out[:max_len]  # tokenizer_model_max_length truncation — vita_arch.py:326-329
    return out
