"""ORACLE-SIDE CPU BASELINE (test infrastructure, not product): the two encoder towers + projector of oracle/encoders.py restated
on torch CPU tensors in fp32 — the same operators the reference's own modules run (nn.Linear, F.layer_norm, softmax, nn.GELU):
multi-threaded BLAS and element-wise kernels instead of numpy's single-threaded exp / erf.  This is what bench.py's
`cpu_baseline` times for the encoder legs (the fp64 numpy checker was 8-14x slower than the reference's
modules on the same cores and made the GPU ratio look better than it is).  Pinned to the fp64 checker by
tests/test_oracle_pin.py::test_torch_encoder_baseline_matches_the_checker.  Reference lines: see oracle/encoders.py, function by function.
Only tests/ and bench.py's cpu_baseline leg may import this module."""
import math

import numpy as np
import torch
import torch.nn.functional as Fn

from oracle import encoders as oe


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def internvit_tower(sd, v, pix):
    """oracle/encoders.py internvit_embeddings + internvit_layer x L + pixel shuffle (modeling_intern_vit.py:107-122, :158-177,
    :213-253; internvit_encoder.py:42-79)."""
    P = "model.vision_tower.vision_tower."
    g = lambda k: _t(sd[P + k])
    n, _, Hh, Ww = pix.shape
    ps, C, nh = v.patch_size, v.hidden_size, v.num_attention_heads
    gh = Hh // ps
    d = C // nh
    x = Fn.conv2d(_t(pix), g("embeddings.patch_embedding.weight").reshape(C, 3, ps, ps), g("embeddings.patch_embedding.bias"), stride=ps)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([g("embeddings.class_embedding").reshape(1, 1, C).expand(n, 1, C), x], 1) + g("embeddings.position_embedding").reshape(1, -1, C)
    N = x.shape[1]
    for l in range(v.num_hidden_layers):
        w = lambda k: g(f"encoder.layers.{l}." + k)
        h = Fn.layer_norm(x, (C,), w("norm1.weight"), w("norm1.bias"), v.layer_norm_eps)
        qkv = Fn.linear(h, w("attn.qkv.weight"), w("attn.qkv.bias")).reshape(n, N, 3, nh, d).permute(2, 0, 3, 1, 4)
        att = torch.softmax((qkv[0] * d ** -0.5) @ qkv[1].transpose(-2, -1), -1)
        a = (att @ qkv[2]).transpose(1, 2).reshape(n, N, C)
        x = x + Fn.linear(a, w("attn.proj.weight"), w("attn.proj.bias")) * w("ls1")
        h = Fn.layer_norm(x, (C,), w("norm2.weight"), w("norm2.bias"), v.layer_norm_eps)
        x = x + Fn.linear(Fn.gelu(Fn.linear(h, w("mlp.fc1.weight"), w("mlp.fc1.bias"))), w("mlp.fc2.weight"), w("mlp.fc2.bias")) * w("ls2")
    f = x[:, 1:].reshape(n, gh, gh, C) * 0.5
    f = f.reshape(n, gh, gh // 2, C * 2).permute(0, 2, 1, 3)
    f = f.reshape(n, gh // 2, gh // 2, C * 4).permute(0, 2, 1, 3)
    return f.reshape(n, -1, C * 4)


def projector(sd, feats):
    g = lambda k: _t(sd["model.mm_projector." + k])
    return Fn.linear(Fn.gelu(Fn.linear(feats, g("0.weight"), g("0.bias"))), g("2.weight"), g("2.bias"))


def whale_encoder(sd, a, feats):
    """oracle/encoders.py whale_encoder for one full-length utterance, full attention (init_model.py:114-139 and the modules it
    calls; no pad / chunk mask: the bench clip has neither)."""
    A = "model.audio_encoder."
    g = lambda k: _t(sd[A + k])
    x = (_t(feats) - g("encoder.global_cmvn.mean")) * g("encoder.global_cmvn.istd")
    c = "encoder.enc.0.core."
    y = torch.relu(Fn.conv2d(x[None, None], g(c + "conv.0.weight"), g(c + "conv.0.bias"), stride=2))
    y = torch.relu(Fn.conv2d(y, g(c + "conv.2.weight"), g(c + "conv.2.bias"), stride=2))[0]
    C, T2, F2 = y.shape
    y = Fn.linear(y.permute(1, 0, 2).reshape(T2, C * F2), g(c + "out.0.weight"), g(c + "out.0.bias"))
    e = "encoder.enc.1."
    y = torch.relu(Fn.layer_norm(Fn.linear(y, g(e + "embed.0.weight"), g(e + "embed.0.bias")), (C,), g(e + "embed.1.weight"),
                                 g(e + "embed.1.bias"), 1e-5)) * math.sqrt(C)
    pos = _t(oe.sinusoid_pe(T2, C))
    nh = a.num_attention_heads
    dk = C // nh
    for l in range(a.num_hidden_layers):
        w = lambda k: g(e + f"encoders.{l}." + k)
        h = Fn.layer_norm(y, (C,), w("norm1.weight"), w("norm1.bias"), a.layer_norm_eps)
        q = Fn.linear(h, w("self_attn.linear_q.weight"), w("self_attn.linear_q.bias")).reshape(T2, nh, dk)
        k = Fn.linear(h, w("self_attn.linear_k.weight"), w("self_attn.linear_k.bias")).reshape(T2, nh, dk)
        vv = Fn.linear(h, w("self_attn.linear_v.weight"), w("self_attn.linear_v.bias")).reshape(T2, nh, dk)
        pp = Fn.linear(pos, w("self_attn.linear_pos.weight")).reshape(T2, nh, dk)
        qu = (q + w("self_attn.pos_bias_u")).transpose(0, 1)
        qv = (q + w("self_attn.pos_bias_v")).transpose(0, 1)
        sc = (qu @ k.permute(1, 2, 0) + qv @ pp.permute(1, 2, 0)) / math.sqrt(dk)
        o = (torch.softmax(sc, -1) @ vv.transpose(0, 1)).transpose(0, 1).reshape(T2, C)
        y = y + Fn.linear(o, w("self_attn.linear_out.weight"), w("self_attn.linear_out.bias"))
        h = Fn.layer_norm(y, (C,), w("norm2.weight"), w("norm2.bias"), a.layer_norm_eps)
        y = y + Fn.linear(torch.relu(Fn.linear(h, w("feed_forward.w_1.weight"), w("feed_forward.w_1.bias"))),
                          w("feed_forward.w_2.weight"), w("feed_forward.w_2.bias"))
    y = Fn.layer_norm(y, (C,), g(e + "after_norm.weight"), g(e + "after_norm.bias"), a.layer_norm_eps)
    kz = a.adapter_kernel
    z = Fn.conv1d(Fn.pad(y.t()[None], (0, kz - 1)), g("adpter.conv1d2.weight"), g("adpter.conv1d2.bias"), stride=2)[0].t()
    z = Fn.gelu(Fn.layer_norm(z, (z.shape[-1],), g("adpter.bn2.weight"), g("adpter.bn2.bias"), a.adapter_norm_eps))
    return Fn.linear(z, g("adpter.project.weight"), g("adpter.project.bias"))
