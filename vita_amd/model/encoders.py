"""HIP implementations of the reference's sub-module plug points:

  InternViTVisionTower   <- vita/model/multimodal_encoder/internvit/internvit_encoder.py:8-106
  VisionProjector        <- vita/model/multimodal_projector/builder.py:154-168 (mlp2x_gelu)
  WhaleAudioEncoder      <- vita/model/multimodal_encoder/whale/init_model.py:63-139

Same constructor-time names, forward signatures and return shapes; the arithmetic is a sequence
of libvita_hip.so kernels (vita_amd.ops), activations fp32, weights bf16.  nn.Module is used only
as the container type the reference's callers expect (.to(), .eval(), attribute access)."""
import math
import os

import numpy as np
import torch
from torch import nn

from .. import ops
from ..audio_frontend import AudioEncoderProcessor
from ..config import AudioConfig, VisionConfig

VIT = "model.vision_tower.vision_tower."
AUD = "model.audio_encoder."


def _get(sd, k):
    v = sd[k]
    return torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v


def _bf(x, device):
    return x.to(device=device, dtype=torch.bfloat16).contiguous()


def _f32(x, device):
    return x.to(device=device, dtype=torch.float32).contiguous()


# VITA_AMD_VIT_PLANES=0: the ViT blocks on the general GEMM with fp32 rows between the operators (the r01-r03 form; A/B and bisecting)
_VIT_PLANES = os.environ.get("VITA_AMD_VIT_PLANES", "1") != "0"

class _HipModule(nn.Module):
    """Container base: weights are plain tensors kept out of nn.Parameter bookkeeping (they are in
    kernel layout, not the checkpoint's), so .to(dtype=...) from the demo is a no-op by design."""

    def __init__(self):
        super().__init__()
        self._device = torch.device("cuda:0")

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return torch.float32  # activations are fp32; see DESIGN.md "numeric contract"

    def to(self, *args, **kwargs):  # dtype moves are ignored: the kernels define the dtypes
        return self

    def half(self):
        return self

    def float(self):
        return self


# =============================================================================================
class InternViTVisionTower(_HipModule):
    def __init__(self, vision_tower=None, args=None, delay_load=False, vcfg: VisionConfig = None):
        super().__init__()
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = -1
        self.scale_pix_shuffle = 0.5
        self.vcfg = vcfg or VisionConfig()
        self.image_processor = None
        self._sd = None
        self.w = None
        # True: one library call per operator (vh_gemm / vh_attention / ...) instead of per block (debugging, A/B timing)
        self.per_operator = os.environ.get("VITA_AMD_PER_OPERATOR", "0") == "1"

    # -- loading ------------------------------------------------------------------------------
    def set_state_dict(self, sd, device):
        self._sd, self._device = sd, torch.device(device)

    def load_model(self):
        if self.is_loaded:
            return
        if self._sd is None:
            raise RuntimeError("InternViTVisionTower.load_model(): no weights attached (set_state_dict first)")
        from ..host.image_processing import make_image_processor
        self.image_processor = make_image_processor(self.vcfg.image_size)
        self.w = self._pack(self._sd, self.vcfg, self._device)
        self.is_loaded = True

    @staticmethod
    def _pack(sd, v, dev):
        C = v.hidden_size
        kk = 3 * v.patch_size * v.patch_size
        kpad = (kk + 63) // 64 * 64
        wp = torch.zeros((C, kpad), dtype=torch.float32)
        wp[:, :kk] = _get(sd, VIT + "embeddings.patch_embedding.weight").reshape(C, kk).float()
        w = {"kpad": kpad, "patch_w": _bf(wp, dev), "patch_b": _f32(_get(sd, VIT + "embeddings.patch_embedding.bias"), dev),
             "cls": _bf(_get(sd, VIT + "embeddings.class_embedding").reshape(C), dev),
             "pos": _bf(_get(sd, VIT + "embeddings.position_embedding").reshape(-1, C), dev), "layers": []}
        for l in range(v.num_hidden_layers):
            p = VIT + f"encoder.layers.{l}."
            g = lambda k: _get(sd, p + k)
            w["layers"].append({
                "n1w": _f32(g("norm1.weight"), dev), "n1b": _f32(g("norm1.bias"), dev),
                "qkv_w": _bf(g("attn.qkv.weight"), dev), "qkv_b": _f32(g("attn.qkv.bias"), dev),
                "proj_w": _bf(g("attn.proj.weight"), dev), "proj_b": _f32(g("attn.proj.bias"), dev),
                "ls1": _f32(g("ls1"), dev),
                "n2w": _f32(g("norm2.weight"), dev), "n2b": _f32(g("norm2.bias"), dev),
                "fc1_w": _bf(g("mlp.fc1.weight"), dev), "fc1_b": _f32(g("mlp.fc1.bias"), dev),
                "fc2_w": _bf(g("mlp.fc2.weight"), dev), "fc2_b": _f32(g("mlp.fc2.bias"), dev),
                "ls2": _f32(g("ls2"), dev)})
        return w

    # -- forward ------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, images, want_layers=False):
        """images [n,3,S,S] (or a list of [3,S,S]) -> [n, (S/28)^2, 4*hidden]  (internvit_encoder.py:55-79)."""
        if not self.is_loaded:
            self.load_model()
        if type(images) is list:
            images = torch.stack([im for im in images], 0)
        if images.ndim != 4:
            raise ValueError(f"wrong pixel_values size: {images.shape}")  # modeling_intern_vit.py:377
        v, w = self.vcfg, self.w
        pix = images.to(device=self._device, dtype=torch.float32).contiguous()
        n, C, nh = pix.shape[0], v.hidden_size, v.num_attention_heads
        if pix.shape[2] != v.image_size or pix.shape[3] != v.image_size:
            raise ValueError(f"InternViT HIP tower is built for {v.image_size}x{v.image_size} tiles, got {tuple(pix.shape)}")
        d = C // nh
        N = v.num_tokens
        g = v.grid
        assert g * g == N - 1  # internvit_encoder.py:72
        layers = []
        if not self.per_operator and w["layers"]:
            # one library call per block (vh_encoder_layer): 24 host calls per pass instead of ~170 — and (r05) ONE call for the
            # embeddings + the first norm1 (vh_vit_embed; the five operator calls left 177 us of host gaps in front of block 0,
            # profiles/r04_encoder_pass_trace.txt); the scratch is looked up BEFORE the first kernel is enqueued
            sc = ops.encoder_scratch(n * N, C, w["layers"][0]["fc1_w"].shape[0], self._device)
            # planes mode (r04): LayerNorm / attention / GELU outputs travel as the bf16 hi/lo planes the weight-streaming GEMM consumes
            # (one-round tilings: qkv / fc1 25 % faster at one tile, 30-36 % on 8-image batches; profiles/r04_enc_sp_sweep.jsonl)
            planes = _VIT_PLANES and C % 64 == 0 and sc.ws.numel() * sc.ws.element_size() >= 4 * n * N * C
            x = torch.empty((n * N, C), dtype=torch.float32, device=self._device)
            L0 = w["layers"][0]
            ops.vit_embed(pix, v.patch_size, w["kpad"], w["patch_w"], w["patch_b"], w["cls"], w["pos"], L0["n1w"], L0["n1b"],
                          v.layer_norm_eps, N, x, sc.h[0] if planes else sc.h[1], sc.h[1] if planes else None)
            h = sc.h[1]
            for li, L in enumerate(w["layers"]):
                nxt = w["layers"][li + 1] if li + 1 < len(w["layers"]) else None
                ops.encoder_layer(x, h, sc.h[li & 1], L, sc, heads=nh, B=n, act="gelu", eps=v.layer_norm_eps,
                                  next_norm=(nxt["n1w"], nxt["n1b"]) if nxt is not None else None, planes=planes)
                h = sc.h[li & 1]
                if want_layers:
                    layers.append(x.clone().view(n, N, C))
            out = ops.vit_pixel_shuffle(x, n, g, C, self.scale_pix_shuffle)
            return (out, layers) if want_layers else out
        patches = ops.vit_patchify(pix, v.patch_size, w["kpad"])
        pe = ops.gemm(patches, w["patch_w"], bias=w["patch_b"])
        x = ops.vit_assemble(pe, w["cls"], w["pos"], n, N, C)                    # [n*N, C]
        # every LayerNorm after the first rides on the Linear that produces its input (ops.gemm(ln=...): the split-K reducer
        # holds whole rows, so the norm costs no launch of its own at one tile)
        h = ops.layernorm(x, w["layers"][0]["n1w"], w["layers"][0]["n1b"], v.layer_norm_eps) if w["layers"] else None
        attn = torch.empty((n * N, C), dtype=torch.float32, device=self._device)
        for li, L in enumerate(w["layers"]):
            qkv = ops.gemm(h, L["qkv_w"], bias=L["qkv_b"])                        # [n*N, 3C] = (three, head, d)
            ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], attn, B=n, Hq=nh, Hkv=nh, Sq=N, Sk=N, d=d, ldq=3 * C,
                          hsq=d, ldk=3 * C, hsk=d, ldv=3 * C, hsv=d, ldo=C, bsq=N * 3 * C, bsk=N * 3 * C, bso=N * C,
                          scale=d ** -0.5)
            _, h = ops.gemm(attn, L["proj_w"], bias=L["proj_b"], scale=L["ls1"], resid=x, out=x,
                            ln=(L["n2w"], L["n2b"], v.layer_norm_eps))
            m = ops.gemm(h, L["fc1_w"], bias=L["fc1_b"], act="gelu")
            nxt = w["layers"][li + 1] if li + 1 < len(w["layers"]) else None
            if nxt is not None:
                _, h = ops.gemm(m, L["fc2_w"], bias=L["fc2_b"], scale=L["ls2"], resid=x, out=x,
                                ln=(nxt["n1w"], nxt["n1b"], v.layer_norm_eps))
            else:
                ops.gemm(m, L["fc2_w"], bias=L["fc2_b"], scale=L["ls2"], resid=x, out=x)
            if want_layers:
                layers.append(x.clone().view(n, N, C))
        out = ops.vit_pixel_shuffle(x, n, g, C, self.scale_pix_shuffle)
        return (out, layers) if want_layers else out

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def config(self):
        return self.vcfg

    @property
    def hidden_size(self):
        return self.vcfg.hidden_size * (int(1 / self.scale_pix_shuffle) ** 2)

    @property
    def num_patches(self):
        return (self.vcfg.image_size // self.vcfg.patch_size) ** 2


# =============================================================================================
class VisionProjector(_HipModule):
    """mlp2x_gelu: Linear(mm_hidden, hidden) -> GELU -> Linear(hidden, hidden)."""

    def __init__(self, sd=None, device="cuda:0"):
        super().__init__()
        self.w = None
        if sd is not None:
            self.load(sd, device)

    def load(self, sd, device):
        self._device = torch.device(device)
        g = lambda k: _get(sd, "model.mm_projector." + k)
        self.w = {"w0": _bf(g("0.weight"), device), "b0": _f32(g("0.bias"), device),
                  "w2": _bf(g("2.weight"), device), "b2": _f32(g("2.bias"), device)}

    @torch.no_grad()
    def forward(self, x):
        shp = x.shape
        x2 = x.to(device=self._device, dtype=torch.float32).reshape(-1, shp[-1]).contiguous()
        h = ops.gemm(x2, self.w["w0"], bias=self.w["b0"], act="gelu")
        y = ops.gemm(h, self.w["w2"], bias=self.w["b2"])
        return y.view(*shp[:-1], y.shape[-1])


# =============================================================================================
def sinusoid_table(length, d):
    """fp32 sin/cos table of whale's PositionalEncoding (attention.py:26-37)."""
    pos = np.arange(0, length, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / d))).astype(np.float32)
    ang = (pos * div).astype(np.float32)
    pe = np.zeros((length, d), np.float32)
    pe[:, 0::2] = np.sin(ang)
    pe[:, 1::2] = np.cos(ang)
    return pe


class WhaleAudioEncoder(_HipModule):
    """encoder (CMVN -> Conv2dSubsampling4 -> 24 rel-pos transformer layers) + CNNSubsampling adapter.
    forward(speech [B,T,80], speech_lengths [B]) -> {"inputs_embeds": [B,T'',H], "attention_mask": [B,T''] bool}.

    The reference draws a RANDOM chunk mask at inference when transformer-dynamic-chunks is on
    (transformer.py:383-384, utils.py:122-131 — SURVEY hazard H1).  That is pinned: full attention
    by default, or an explicit deterministic (chunk, left) mask via set_chunk_mask()."""

    def __init__(self, sd=None, acfg: AudioConfig = None, device="cuda:0", llm_dim=4096):
        super().__init__()
        self.acfg = acfg or AudioConfig()
        self.llm_dim = llm_dim
        self.audio_processor = AudioEncoderProcessor()
        self.chunk, self.left = 0, -1
        self.normalized_input = False   # True: features already CMVN-normalised (vLLM-flavour extractor)
        self.per_operator = os.environ.get("VITA_AMD_PER_OPERATOR", "0") == "1"   # one library call per operator, not per block
        self.w = None
        if sd is not None:
            self.load(sd, device)

    def set_chunk_mask(self, chunk, left=-1):
        self.chunk, self.left = int(chunk), int(left)

    def load(self, sd, device):
        self._device = dev = torch.device(device)
        a = self.acfg
        C, Fq = a.hidden_size, a.sub_freq
        g = lambda k: _get(sd, AUD + k)
        c = "encoder.enc.0.core."
        has_cmvn = (AUD + "encoder.global_cmvn.mean") in sd
        self.cmvn_from_checkpoint = has_cmvn
        if has_cmvn:
            mean, istd = g("encoder.global_cmvn.mean"), g("encoder.global_cmvn.istd")
        else:  # the HF-path checkpoint keeps CMVN in <mm_audio_encoder>/global_cmvn (loaded by the builder)
            mean, istd = torch.zeros(a.input_dim), torch.ones(a.input_dim)
        w = {"mean": _f32(mean, dev), "istd": _f32(istd, dev),
             "c1_w": _bf(g(c + "conv.0.weight").reshape(C, 9), dev), "c1_b": _f32(g(c + "conv.0.bias"), dev),
             # [Cout, Cin, kh, kw] -> [Cout, (kh, kw, Cin)]: each (kh,kw) is a contiguous channel run
             "c2_w": _bf(g(c + "conv.2.weight").permute(0, 2, 3, 1).reshape(C, 9 * C), dev),
             "c2_b": _f32(g(c + "conv.2.bias"), dev),
             # Linear over (c, f) flattened c-major -> columns reordered to (f, c) to match [T', F', C] rows
             "out_w": _bf(g(c + "out.0.weight").reshape(C, C, Fq).permute(0, 2, 1).reshape(C, Fq * C), dev),
             "out_b": _f32(g(c + "out.0.bias"), dev)}
        e = "encoder.enc.1."
        w.update({"emb_w": _bf(g(e + "embed.0.weight"), dev), "emb_b": _f32(g(e + "embed.0.bias"), dev),
                  "emb_nw": _f32(g(e + "embed.1.weight"), dev), "emb_nb": _f32(g(e + "embed.1.bias"), dev),
                  "an_w": _f32(g(e + "after_norm.weight"), dev), "an_b": _f32(g(e + "after_norm.bias"), dev),
                  "layers": []})
        for l in range(a.num_hidden_layers):
            p = e + f"encoders.{l}."
            q = lambda k: g(p + k)
            w["layers"].append({
                "n1w": _f32(q("norm1.weight"), dev), "n1b": _f32(q("norm1.bias"), dev),
                "qkv_w": _bf(torch.cat([q("self_attn.linear_q.weight"), q("self_attn.linear_k.weight"),
                                        q("self_attn.linear_v.weight")], 0), dev),
                "qkv_b": _f32(torch.cat([q("self_attn.linear_q.bias"), q("self_attn.linear_k.bias"),
                                         q("self_attn.linear_v.bias")], 0), dev),
                "pos_w": _bf(q("self_attn.linear_pos.weight"), dev),
                "u": _f32(q("self_attn.pos_bias_u"), dev), "v": _f32(q("self_attn.pos_bias_v"), dev),
                "out_w": _bf(q("self_attn.linear_out.weight"), dev), "out_b": _f32(q("self_attn.linear_out.bias"), dev),
                "n2w": _f32(q("norm2.weight"), dev), "n2b": _f32(q("norm2.bias"), dev),
                "w1": _bf(q("feed_forward.w_1.weight"), dev), "b1": _f32(q("feed_forward.w_1.bias"), dev),
                "w2": _bf(q("feed_forward.w_2.weight"), dev), "b2": _f32(q("feed_forward.w_2.bias"), dev)})
            Ld = w["layers"][-1]    # the block entry's names (ops.encoder_layer)
            Ld.update(proj_w=Ld["out_w"], proj_b=Ld["out_b"], fc1_w=Ld["w1"], fc1_b=Ld["b1"], fc2_w=Ld["w2"], fc2_b=Ld["b2"])
        ad = "adpter."
        k = a.adapter_kernel
        w.update({"ad_w": _bf(g(ad + "conv1d2.weight").permute(0, 2, 1).reshape(2 * C, k * C), dev),
                  "ad_b": _f32(g(ad + "conv1d2.bias"), dev),
                  "ad_nw": _f32(g(ad + "bn2.weight"), dev), "ad_nb": _f32(g(ad + "bn2.bias"), dev),
                  "pj_w": _bf(g(ad + "project.weight"), dev), "pj_b": _f32(g(ad + "project.bias"), dev)})
        self.pe = torch.from_numpy(sinusoid_table(a.max_pe_len, C)).to(dev)
        self.w = w
        self._idx_cache = {}
        self._pos_proj = None       # [layers][cap, C]: linear_pos(pos_emb) of every layer for the first `cap` positions

    def _pos_projections(self, T2):
        """linear_pos(pos_emb[:T2]) of all layers.  The operand is the fixed sinusoid table (attention.py:26-37, :358-369: `p =
        self.linear_pos(pos_emb)`), not the audio, so it is a function of the weights alone — folded once per table length
        (grown in powers of two, rows [0, T2) of a longer fold are the fold of pos_emb[:T2]) like a RoPE table, instead of 24
        GEMMs per clip.  100 MB at 1024 positions (40 s of audio)."""
        cap = 0 if self._pos_proj is None else self._pos_proj[0].shape[0]
        if T2 > cap:
            cap = min(max(256, 1 << (T2 - 1).bit_length()), self.acfg.max_pe_len)
            if T2 > cap:
                raise ValueError(f"audio clip needs {T2} positions, the table holds {self.acfg.max_pe_len}")
            pos = self.pe[:cap]
            self._pos_proj = [ops.gemm(pos, L["pos_w"]) for L in self.w["layers"]]
        return self._pos_proj

    def _conv2_rows(self, T1, F1):
        key = ("c2", T1, F1)
        if key not in self._idx_cache:
            T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
            t, f = np.meshgrid(np.arange(T2), np.arange(F2), indexing="ij")
            rows = ((2 * t) * F1 + 2 * f).reshape(-1).astype(np.int32)
            self._idx_cache[key] = (torch.from_numpy(rows).to(self._device), T2, F2,
                                    [kh * F1 + kw for kh in range(3) for kw in range(3)])
        return self._idx_cache[key]

    def _adapter_rows(self, T2):
        key = ("ad", T2)
        if key not in self._idx_cache:
            k = self.acfg.adapter_kernel
            T3 = (T2 + k - 1 - k) // 2 + 1
            self._idx_cache[key] = (torch.from_numpy((2 * np.arange(T3)).astype(np.int32)).to(self._device), T3)
        return self._idx_cache[key]

    @torch.no_grad()
    def encode_one(self, feats, length=None, want_layers=False, dbg=None):
        """feats fp32 [T, 80] on device, `length` valid frames.  Returns (embeds [T'', H], mask [T''] bool)."""
        a, w = self.acfg, self.w
        C, nh = a.hidden_size, a.num_attention_heads
        dk = C // nh
        feats = feats.to(device=self._device, dtype=torch.float32).contiguous()
        T = feats.shape[0]
        length = T if length is None else int(length)
        if self.normalized_input:
            if "mean0" not in w:
                w["mean0"], w["istd1"] = torch.zeros_like(w["mean"]), torch.ones_like(w["istd"])
            mean, istd = w["mean0"], w["istd1"]
        else:
            mean, istd = w["mean"], w["istd"]
        y1, T1, F1 = ops.audio_conv1(feats, mean, istd, w["c1_w"], w["c1_b"])                  # [T1*F1, C]
        rows, T2, F2, segrow = self._conv2_rows(T1, F1)
        y2 = ops.gemm(y1, w["c2_w"], bias=w["c2_b"], act="relu", a_rowidx=rows, segrow=segrow, seglen=C)
        y = ops.gemm(y2.view(T2, F2 * C), w["out_w"], bias=w["out_b"])                          # [T2, C]
        if dbg is not None:
            dbg.update(conv1=y1.view(T1, F1, C).clone(), conv2=y2.view(T2, F2, C).clone(), sub_out=y.clone())
        klen = len(range(T)[:length][2::2][2::2])                                               # x_mask[:, :, 2::2][:, :, 2::2]
        y = ops.gemm(y, w["emb_w"], bias=w["emb_b"])
        y = ops.layernorm(y, w["emb_nw"], w["emb_nb"], 1e-5, act="relu", post_scale=math.sqrt(C))
        if dbg is not None:
            dbg["embed"] = y.clone()
        pos_proj = self._pos_projections(T2)
        o = torch.empty((T2, C), dtype=torch.float32, device=self._device)
        layers = []
        h = ops.layernorm(y, w["layers"][0]["n1w"], w["layers"][0]["n1b"], a.layer_norm_eps) if w["layers"] else None
        block_calls = not self.per_operator and bool(w["layers"])
        if block_calls:
            sc = ops.encoder_scratch(T2, C, w["layers"][0]["fc1_w"].shape[0], self._device)
            for li, L in enumerate(w["layers"]):
                nxt = w["layers"][li + 1] if li + 1 < len(w["layers"]) else None
                nn_ = (nxt["n1w"], nxt["n1b"]) if nxt is not None else (w["an_w"], w["an_b"])
                ops.encoder_layer(y, h, sc.h[li & 1], L, sc, heads=nh, B=1, act="relu", eps=a.layer_norm_eps,
                                  next_norm=nn_, p=pos_proj[li], bias_u=L["u"], bias_v=L["v"], klen=klen, chunk=self.chunk,
                                  left=self.left)
                h = sc.h[li & 1]
                if want_layers:
                    layers.append(y.clone())
        for li, L in enumerate(w["layers"] if not block_calls else []):
            qkv = ops.gemm(h, L["qkv_w"], bias=L["qkv_b"])
            pp = pos_proj[li]
            ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], o, B=1, Hq=nh, Hkv=nh, Sq=T2, Sk=T2, d=dk, ldq=3 * C, hsq=dk,
                          ldk=3 * C, hsk=dk, ldv=3 * C, hsv=dk, ldo=C, scale=1.0 / math.sqrt(dk), klen=klen,
                          chunk=self.chunk, left=self.left, p=pp, ldp=C, hsp=dk, bias_u=L["u"], bias_v=L["v"])
            _, h = ops.gemm(o, L["out_w"], bias=L["out_b"], resid=y, out=y, ln=(L["n2w"], L["n2b"], a.layer_norm_eps))
            f = ops.gemm(h, L["w1"], bias=L["b1"], act="relu")
            # the norm after this layer: the next layer's norm1, or the encoder's after_norm
            nxt = w["layers"][li + 1] if li + 1 < len(w["layers"]) else None
            nw, nb = (nxt["n1w"], nxt["n1b"]) if nxt is not None else (w["an_w"], w["an_b"])
            _, h = ops.gemm(f, L["w2"], bias=L["b2"], resid=y, out=y, ln=(nw, nb, a.layer_norm_eps))
            if want_layers:
                layers.append(y.clone())
        y = h if w["layers"] else ops.layernorm(y, w["an_w"], w["an_b"], a.layer_norm_eps)
        if klen < T2:
            y[klen:].zero_()                                                                    # masked_fill_(~mask_pad, 0)
        rows3, T3 = self._adapter_rows(T2)
        z = ops.gemm(y, w["ad_w"], bias=w["ad_b"], a_rowidx=rows3, segrow=list(range(a.adapter_kernel)), seglen=C)
        z = ops.layernorm(z, w["ad_nw"], w["ad_nb"], a.adapter_norm_eps, act="gelu")
        out = ops.gemm(z, w["pj_w"], bias=w["pj_b"])
        mask = torch.zeros(T2, dtype=torch.bool, device=self._device)
        mask[:klen] = True
        mask = mask[0::2]
        assert out.shape[0] == mask.shape[0]                                                    # init_model.py:127
        return (out, mask, layers) if want_layers else (out, mask)

    @torch.no_grad()
    def forward(self, speech, speech_lengths):
        speech = speech.to(device=self._device, dtype=torch.float32)
        if speech.ndim != 3:
            raise AssertionError("speech must be [B, T, D]")                                   # encoder.py:137
        lens = [int(round(float(x))) for x in speech_lengths.reshape(-1).tolist()]
        embs, masks = [], []
        for b in range(speech.shape[0]):
            e, m = self.encode_one(speech[b].contiguous(), lens[b])
            embs.append(e); masks.append(m)
        return {"inputs_embeds": torch.stack(embs, 0), "attention_mask": torch.stack(masks, 0)}
