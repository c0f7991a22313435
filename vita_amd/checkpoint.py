"""Checkpoint handling: the reference's parameter names (HF transformers 4.41.1 Mixtral + the
VITA towers; name map in web_demo/vllm_tools/vllm_file/mixtral.py:1197-1229) ->
the packed device layout the HIP kernels stream.

No real checkpoint exists offline, so `synth_state_dict` makes a deterministic random one with
exactly those names; values are bf16-representable so the fp32 oracle and the bf16 device copy
hold identical numbers.
"""
import math

import numpy as np

from .config import VitaConfig

LLM = "model.layers.{}."
VIT = "model.vision_tower.vision_tower."
AUD = "model.audio_encoder."


def round_bf16(x):
    """float32 -> nearest-even bf16, returned as float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def synth_state_dict(cfg: VitaConfig, seed=0, rich=True, parts=("text", "vision", "audio")):
    """name -> float32 ndarray.  rich=True perturbs norm weights / biases / layer-scales so a kernel
    that drops one of them fails parity; rich=False is SURVEY §8(d)'s plain init (N(0,0.02),
    norms 1, biases 0, ls 1)."""
    rng = np.random.default_rng(seed)
    sd = {}

    def W(*shape, std=0.02):
        return round_bf16(rng.standard_normal(shape, dtype=np.float32) * std)

    def ones(n):
        if not rich:
            return np.ones(n, np.float32)
        return round_bf16(1.0 + rng.standard_normal(n, dtype=np.float32) * 0.1)

    def bias(n):
        return W(n) if rich else np.zeros(n, np.float32)

    if "text" in parts:
        t = cfg.text
        H, I, E, hd = t.hidden_size, t.intermediate_size, t.num_local_experts, t.head_dim
        sd["model.embed_tokens.weight"] = W(t.vocab_size, H, std=1.0 if rich else 0.02)  # rich: token identity dominates
        for l in range(t.num_hidden_layers):
            p = LLM.format(l)
            sd[p + "input_layernorm.weight"] = ones(H)
            sd[p + "self_attn.q_proj.weight"] = W(t.num_attention_heads * hd, H)
            sd[p + "self_attn.k_proj.weight"] = W(t.num_key_value_heads * hd, H)
            sd[p + "self_attn.v_proj.weight"] = W(t.num_key_value_heads * hd, H)
            sd[p + "self_attn.o_proj.weight"] = W(H, t.num_attention_heads * hd)
            sd[p + "post_attention_layernorm.weight"] = ones(H)
            sd[p + "block_sparse_moe.gate.weight"] = W(E, H, std=0.02 if not rich else 0.08)
            for e in range(E):
                q = p + f"block_sparse_moe.experts.{e}."
                sd[q + "w1.weight"] = W(I, H)
                sd[q + "w2.weight"] = W(H, I)
                sd[q + "w3.weight"] = W(I, H)
        sd["model.norm.weight"] = ones(H)
        sd["lm_head.weight"] = W(t.vocab_size, H)

    if "vision" in parts:
        v = cfg.vision
        C, M = v.hidden_size, v.intermediate_size
        sd[VIT + "embeddings.class_embedding"] = W(1, 1, C, std=0.5 if rich else 1.0)
        sd[VIT + "embeddings.patch_embedding.weight"] = W(C, 3, v.patch_size, v.patch_size)
        sd[VIT + "embeddings.patch_embedding.bias"] = bias(C)
        sd[VIT + "embeddings.position_embedding"] = W(1, v.num_tokens, C, std=0.5 if rich else 1.0)
        for l in range(v.num_hidden_layers):
            p = VIT + f"encoder.layers.{l}."
            sd[p + "norm1.weight"] = ones(C); sd[p + "norm1.bias"] = bias(C)
            sd[p + "attn.qkv.weight"] = W(3 * C, C); sd[p + "attn.qkv.bias"] = bias(3 * C)
            sd[p + "attn.proj.weight"] = W(C, C); sd[p + "attn.proj.bias"] = bias(C)
            sd[p + "ls1"] = ones(C)
            sd[p + "norm2.weight"] = ones(C); sd[p + "norm2.bias"] = bias(C)
            sd[p + "mlp.fc1.weight"] = W(M, C); sd[p + "mlp.fc1.bias"] = bias(M)
            sd[p + "mlp.fc2.weight"] = W(C, M); sd[p + "mlp.fc2.bias"] = bias(C)
            sd[p + "ls2"] = ones(C)
        Ht = cfg.text.hidden_size
        sd["model.mm_projector.0.weight"] = W(Ht, v.out_dim); sd["model.mm_projector.0.bias"] = bias(Ht)
        sd["model.mm_projector.2.weight"] = W(Ht, Ht); sd["model.mm_projector.2.bias"] = bias(Ht)

    if "audio" in parts:
        a = cfg.audio
        C, M, F = a.hidden_size, a.intermediate_size, a.sub_freq
        nh = a.num_attention_heads
        dk = C // nh
        mean, istd = vendored_cmvn(a.input_dim)
        sd[AUD + "encoder.global_cmvn.mean"] = mean
        sd[AUD + "encoder.global_cmvn.istd"] = istd
        e0 = AUD + "encoder.enc.0.core."
        sd[e0 + "conv.0.weight"] = W(C, 1, 3, 3, std=0.1); sd[e0 + "conv.0.bias"] = bias(C)
        sd[e0 + "conv.2.weight"] = W(C, C, 3, 3); sd[e0 + "conv.2.bias"] = bias(C)
        sd[e0 + "out.0.weight"] = W(C, C * F); sd[e0 + "out.0.bias"] = bias(C)
        e1 = AUD + "encoder.enc.1."
        sd[e1 + "embed.0.weight"] = W(C, C); sd[e1 + "embed.0.bias"] = bias(C)
        sd[e1 + "embed.1.weight"] = ones(C); sd[e1 + "embed.1.bias"] = bias(C)
        lim = math.sqrt(6.0 / (nh + dk))  # xavier_uniform_ on [h, d_k] (attention.py:313-314)
        for l in range(a.num_hidden_layers):
            p = e1 + f"encoders.{l}."
            sd[p + "norm1.weight"] = ones(C); sd[p + "norm1.bias"] = bias(C)
            for nm in ("linear_q", "linear_k", "linear_v", "linear_out"):
                sd[p + f"self_attn.{nm}.weight"] = W(C, C); sd[p + f"self_attn.{nm}.bias"] = bias(C)
            sd[p + "self_attn.linear_pos.weight"] = W(C, C)
            sd[p + "self_attn.pos_bias_u"] = round_bf16(rng.uniform(-lim, lim, (nh, dk)).astype(np.float32))
            sd[p + "self_attn.pos_bias_v"] = round_bf16(rng.uniform(-lim, lim, (nh, dk)).astype(np.float32))
            sd[p + "norm2.weight"] = ones(C); sd[p + "norm2.bias"] = bias(C)
            sd[p + "feed_forward.w_1.weight"] = W(M, C); sd[p + "feed_forward.w_1.bias"] = bias(M)
            sd[p + "feed_forward.w_2.weight"] = W(C, M); sd[p + "feed_forward.w_2.bias"] = bias(C)
        sd[e1 + "after_norm.weight"] = ones(C); sd[e1 + "after_norm.bias"] = bias(C)
        ad = AUD + "adpter."
        Ht = cfg.text.hidden_size
        sd[ad + "conv1d2.weight"] = W(2 * C, C, a.adapter_kernel); sd[ad + "conv1d2.bias"] = bias(2 * C)
        sd[ad + "bn2.weight"] = ones(2 * C); sd[ad + "bn2.bias"] = bias(2 * C)
        sd[ad + "project.weight"] = W(Ht, 2 * C); sd[ad + "project.bias"] = bias(Ht)
    return sd


_CMVN = None


def vendored_cmvn(dim=80):
    """The 80-dim CMVN means / inverse stds the reference vendors in
    web_demo/vllm_tools/model_weight_file/feature_extractor/preprocessor_config.json; we keep a copy of
    the numbers (data, not code) in vita_amd/data/cmvn.json so they travel to the GPU box."""
    global _CMVN
    if _CMVN is None:
        import json
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cmvn.json")
        with open(path) as f:
            d = json.load(f)
        _CMVN = (np.asarray(d["cmvn_means"], np.float32), np.asarray(d["cmvn_istds"], np.float32))
    mean, istd = _CMVN
    if dim != mean.shape[0]:
        raise ValueError(f"CMVN is {mean.shape[0]}-dimensional, model wants {dim}")
    return mean.copy(), istd.copy()


# ---------------------------------------------------------------------------------------------
# packing for the device
# ---------------------------------------------------------------------------------------------
def _t(x, device, dtype):
    import torch
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=device, dtype=dtype).contiguous()


def tp_slices(cfg_text, rank, world):
    """Tensor-parallel partition (vllm_file/mixtral.py:375-414,441-476): q/kv heads
    column-sharded, every expert on every rank with intermediate/world columns."""
    nq, nkv, I = cfg_text.num_attention_heads, cfg_text.num_key_value_heads, cfg_text.intermediate_size
    if nq % world or nkv % world or I % world or (I // world) % 64:
        raise ValueError(f"tensor-parallel degree {world} does not divide heads {nq}/{nkv} or FFN {I}")
    hd = cfg_text.head_dim
    q = slice(rank * nq // world * hd, (rank + 1) * nq // world * hd)
    kv = slice(rank * nkv // world * hd, (rank + 1) * nkv // world * hd)
    ff = slice(rank * I // world, (rank + 1) * I // world)
    return q, kv, ff


def vocab_shard(vocab, rank, world):
    """rows of the LM head on this rank: ceil(V / world) consecutive rows, the last rank takes the remainder
    (ParallelLMHead of the reference's vLLM flavour, vllm_file/mixtral.py:939-951; no padding rows: the candidate
    exchange carries global indices)."""
    if world <= 1:
        return 0, vocab
    per = -(-vocab // world)
    lo = min(rank * per, vocab)
    return lo, max(0, min(per, vocab - lo))


def pack_mixtral(sd, cfg: VitaConfig, device, rank=0, world=1, shard_vocab=True):
    """state dict (reference names) -> dict of device tensors in engine layout for this TP rank."""
    import torch
    t = cfg.text
    qs, kvs, ff = tp_slices(t, rank, world)
    bf, f32 = torch.bfloat16, torch.float32
    g = lambda k: sd[k]
    lo, n = vocab_shard(t.vocab_size, rank, world if shard_vocab else 1)
    out = {"embed": _t(g("model.embed_tokens.weight"), device, bf),
           "final_norm": _t(g("model.norm.weight"), device, f32),
           "lm_head": _t(g("lm_head.weight")[lo:lo + n], device, bf), "layers": []}
    if world > 1 and shard_vocab:
        out["vocab_lo"], out["vocab_n"] = lo, n
    for l in range(t.num_hidden_layers):
        p = LLM.format(l)
        cat = np.concatenate if isinstance(g(p + "self_attn.q_proj.weight"), np.ndarray) else torch.cat
        stack = np.stack if cat is np.concatenate else torch.stack
        wqkv = cat([g(p + "self_attn.q_proj.weight")[qs], g(p + "self_attn.k_proj.weight")[kvs],
                    g(p + "self_attn.v_proj.weight")[kvs]], 0)
        ex = lambda nm: [g(p + f"block_sparse_moe.experts.{e}.{nm}.weight") for e in range(t.num_local_experts)]
        layer = {
            "attn_norm": _t(g(p + "input_layernorm.weight"), device, f32),
            "wqkv": _t(wqkv, device, bf),
            "wo": _t(g(p + "self_attn.o_proj.weight")[:, qs], device, bf),
            "ffn_norm": _t(g(p + "post_attention_layernorm.weight"), device, f32),
            "wrouter": _t(g(p + "block_sparse_moe.gate.weight"), device, bf),
            "w1": _t(stack([w[ff] for w in ex("w1")], 0), device, bf),
            "w3": _t(stack([w[ff] for w in ex("w3")], 0), device, bf),
            "w2": _t(stack([w[:, ff] for w in ex("w2")], 0), device, bf),
        }
        out["layers"].append(layer)
    return out


def tensor_seed(name, base=0):
    """64-bit stream id of a reference-named tensor (FNV-1a over the name, mixed with the run's seed) for the
    counter-based weight generator vh_fill_hash_bf16; oracle/hashw.py computes the same id on the host."""
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (int(base) * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def hash_fill(dst, name, seed=0, ld_src=None, idx0=0):
    """dst: bf16 device tensor view [rows, cols] (rows may be strided) <- the values of reference tensor `name`."""
    import ctypes as C

    import torch

    from . import _lib
    if dst.dtype != torch.bfloat16 or dst.ndim != 2 or dst.stride(1) != 1:
        raise ValueError("hash_fill wants a 2-D bfloat16 view contiguous along its last dimension")
    rows, cols = dst.shape
    _lib.check(_lib.load().vh_fill_hash_bf16(C.c_void_p(dst.data_ptr()), rows, cols, dst.stride(0),
                                             cols if ld_src is None else int(ld_src), int(idx0),
                                             C.c_uint64(tensor_seed(name, seed)),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vh_fill_hash_bf16")
    return dst


def synth_mixtral_device(cfg: VitaConfig, device, seed=0, rank=0, world=1, shard_vocab=True):
    """Random-init backbone generated directly on the GPU in packed layout (the full model is 93.7 GB bf16, too
    large to stage through host numpy).  SURVEY §8(d) init (sigma ~0.02, norms 1) from the counter-based generator
    vh_fill_hash_bf16, keyed by the REFERENCE'S parameter names: a tensor-parallel rank's shard holds exactly the
    values of its slice of the full tensor, and the CPU oracle regenerates any tensor on the host
    (oracle/hashw.py) — the released geometry can be parity-checked without a checkpoint."""
    import torch
    t = cfg.text
    qs, kvs, ff = tp_slices(t, rank, world)
    nq = (qs.stop - qs.start) // t.head_dim
    nkv = (kvs.stop - kvs.start) // t.head_dim
    I = ff.stop - ff.start
    H, E, hd = t.hidden_size, t.num_local_experts, t.head_dim
    bf = torch.bfloat16
    ones = lambda n: torch.ones(n, device=device, dtype=torch.float32)
    new = lambda *shape: torch.empty(shape, device=device, dtype=bf)
    lo, nv = vocab_shard(t.vocab_size, rank, world if shard_vocab else 1)
    out = {"embed": hash_fill(new(t.vocab_size, H), "model.embed_tokens.weight", seed), "final_norm": ones(H),
           "lm_head": hash_fill(new(nv, H), "lm_head.weight", seed, idx0=lo * H), "layers": []}
    if world > 1 and shard_vocab:
        out["vocab_lo"], out["vocab_n"] = lo, nv
    for l in range(t.num_hidden_layers):
        p = LLM.format(l)
        wqkv = new((nq + 2 * nkv) * hd, H)
        hash_fill(wqkv[:nq * hd], p + "self_attn.q_proj.weight", seed, idx0=qs.start * H)
        hash_fill(wqkv[nq * hd:(nq + nkv) * hd], p + "self_attn.k_proj.weight", seed, idx0=kvs.start * H)
        hash_fill(wqkv[(nq + nkv) * hd:], p + "self_attn.v_proj.weight", seed, idx0=kvs.start * H)
        wo = hash_fill(new(H, nq * hd), p + "self_attn.o_proj.weight", seed, ld_src=t.num_attention_heads * hd,
                       idx0=qs.start)
        w1, w3, w2 = new(E, I, H), new(E, I, H), new(E, H, I)
        for e in range(E):
            q = p + f"block_sparse_moe.experts.{e}."
            hash_fill(w1[e], q + "w1.weight", seed, idx0=ff.start * H)
            hash_fill(w3[e], q + "w3.weight", seed, idx0=ff.start * H)
            hash_fill(w2[e], q + "w2.weight", seed, ld_src=t.intermediate_size, idx0=ff.start)
        out["layers"].append({
            "attn_norm": ones(H), "wqkv": wqkv, "wo": wo, "ffn_norm": ones(H),
            "wrouter": hash_fill(new(E, H), p + "block_sparse_moe.gate.weight", seed), "w1": w1, "w3": w3, "w2": w2})
    return out
