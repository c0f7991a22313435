"""Thin torch-tensor front end over the C ABI.  torch is plumbing here (device memory, current
stream); every computation is a HIP kernel in libvita_hip.so.  All operators require CUDA
(ROCm) tensors and raise otherwise — there is deliberately no CPU path."""
import ctypes as C

import torch

from . import _lib
from ._lib import AttnArgs, GemmArgs, check

ACT = {None: 0, "none": 0, "gelu": 1, "relu": 2, "silu": 3}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.VitaHipError("vita_amd operators need GPU tensors (no CPU fallback)")


def _c(t, name="tensor"):
    """kernels index raw pointers: require a dense row-major tensor (rows may be strided views)."""
    if t is not None and t.ndim >= 1 and t.stride(-1) != 1 and t.shape[-1] != 1:
        raise ValueError(f"{name} must be contiguous along its last dimension (got strides {t.stride()})")
    return t


def _dense(t):
    return t if t.is_contiguous() else t.contiguous()


def _f32(t, name="tensor"):
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t


def _bf16(t, name="weight"):
    if t.dtype != torch.bfloat16:
        raise TypeError(f"{name} must be bfloat16, got {t.dtype}")
    return t


_SCRATCH = {}


def _scratch(device, nbytes):
    """per-device scratch for split-K partial sums (stream-ordered reuse: every GEMM here runs on the current stream)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _SCRATCH[key] = t
    return t


def gemm(a, w, *, w_up=None, bias=None, act=None, scale=None, resid=None, out=None, a_rowidx=None, a_rows=None,
         segrow=None, seglen=None, m=None, c_rowidx=None, group_off=None, ngroups=0, w_group_stride=0, lda=None,
         ldc=None, ksplit=0, ln=None):
    """out[orow(m), :] = epilogue(A[arow(m, k)] @ w.T).  a: fp32 [rows, lda]; w: bf16 [N, K] (or [E, N, K] grouped).
    ksplit: 0 = the library may split K over more blocks for small launches (plain GEMMs), 1 = never, n = n-way.
    ln = (weight, bias or None, eps): also returns LayerNorm(out) — (out, ln_out) — from the same call (vh_gemm_ln)."""
    _dev(a, w, out)
    _f32(_c(a, "a"), "a"); _bf16(w, "w")
    if not w.is_contiguous():
        raise ValueError("w must be contiguous")
    lib = _lib.load()
    g = GemmArgs()
    N, K = int(w.shape[-2]), int(w.shape[-1])
    g.A = a.data_ptr(); g.lda = int(lda if lda is not None else a.stride(0))
    g.a_rows = int(a_rows if a_rows is not None else a.shape[0])
    g.a_rowidx = a_rowidx.data_ptr() if a_rowidx is not None else None
    if segrow is None:
        g.nseg, g.seglen = 1, K
        g.segrow[0] = 0
    else:
        g.nseg, g.seglen = len(segrow), int(seglen)
        for i, v in enumerate(segrow):
            g.segrow[i] = int(v)
    g.W = w.data_ptr(); g.W_up = w_up.data_ptr() if w_up is not None else None
    g.ldw = K; g.w_group_stride = int(w_group_stride)
    g.group_off = group_off.data_ptr() if group_off is not None else None
    g.ngroups = int(ngroups)
    M = int(m if m is not None else (a_rowidx.shape[0] if a_rowidx is not None else a.shape[0]))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    g.C = out.data_ptr(); g.ldc = int(ldc if ldc is not None else out.stride(0))
    g.c_rowidx = c_rowidx.data_ptr() if c_rowidx is not None else None
    g.bias = bias.data_ptr() if bias is not None else None
    g.scale = scale.data_ptr() if scale is not None else None
    if resid is not None:
        g.resid = resid.data_ptr(); g.ldr = int(resid.stride(0))
    g.M, g.N, g.K, g.act = M, N, K, ACT[act]
    g.ksplit = int(ksplit)
    if ksplit != 1 and group_off is None and w_up is None and c_rowidx is None and M * N <= (8 << 20):
        # room for up to 8 partial-sum slabs, capped at 96 MB (the library lowers the split to what fits)
        ws = _scratch(a.device, min(8 * 4 * M * N, 96 << 20))
        g.ws, g.ws_bytes = ws.data_ptr(), ws.numel()
    if ln is not None:
        lw, lb, eps = ln
        _dev(lw); _f32(_c(lw, "ln weight"), "ln weight")
        ln_out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        check(lib.vh_gemm_ln(C.byref(g), _p(lw), _p(lb), float(eps), _p(ln_out), int(ln_out.stride(0)), _stream()), "vh_gemm_ln")
        return out, ln_out
    check(lib.vh_gemm(C.byref(g), _stream()), "vh_gemm")
    return out


def attention(q, k, v, out, *, B, Hq, Hkv, Sq, Sk, d, ldq, hsq, ldk, hsk, ldv, hsv, ldo, bsq=0, bsk=0, bso=0, scale,
              causal=False, q_off=0, klen=None, chunk=0, left=-1, p=None, ldp=0, hsp=0, bias_u=None, bias_v=None):
    _dev(q, k, v, out)
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (p, "p")):
        _c(t, nm)
    lib = _lib.load()
    a = AttnArgs()
    a.Q, a.ldq, a.hsq = q.data_ptr(), ldq, hsq
    a.K, a.ldk, a.hsk = k.data_ptr(), ldk, hsk
    a.V, a.ldv, a.hsv = v.data_ptr(), ldv, hsv
    if p is not None:
        a.P, a.ldp, a.hsp = p.data_ptr(), ldp, hsp
        a.bias_u, a.bias_v = bias_u.data_ptr(), bias_v.data_ptr()
    a.O, a.ldo = out.data_ptr(), ldo
    a.bsq, a.bsk, a.bso = bsq, bsk, bso
    a.B, a.Hq, a.Hkv, a.Sq, a.Sk, a.d = B, Hq, Hkv, Sq, Sk, d
    a.causal, a.q_off = int(causal), q_off
    a.klen = Sk if klen is None else int(klen)
    a.chunk, a.left = int(chunk), int(left)
    a.scale = float(scale)
    check(lib.vh_attention(C.byref(a), _stream()), "vh_attention")
    return out


class EncoderScratch:
    """Caller-owned scratch of vh_encoder_layer for [M, C] rows and an MLP of width F (reused by every layer of a pass)."""

    def __init__(self, M, Cw, F, device):
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=device)
        self.M, self.C, self.F = M, Cw, F
        self.qkv, self.attn, self.hmid, self.mid = f(M, 3 * Cw), f(M, Cw), f(M, Cw), f(M, F)
        self.h = [f(M, Cw), f(M, Cw)]                       # LayerNorm outputs, ping-pong between layers
        self.ws = _scratch(device, min(8 * 4 * M * max(3 * Cw, F), 96 << 20))


_ENC_SCRATCH = {}
ENC_SCRATCH_CAP_BYTES = 768 << 20      # all cached sets together (an 8-tile ViT batch is ~370 MB, 13 tiles ~600 MB)


def _enc_scratch_bytes(sc):
    return 4 * sc.M * (3 * sc.C + 2 * sc.C + sc.F + 2 * sc.C)


def encoder_scratch(M, Cw, F, device):
    """EncoderScratch for (M, C, F), kept per (device, current stream) across passes: building it costs six allocations (~0.1 ms of host
    time in front of a tower pass, r04 trace); reuse is stream-ordered like _scratch.  At most four shapes and at most
    ENC_SCRATCH_CAP_BYTES in all are kept (dynamic tiling makes M vary per request, and four 13-tile sets would pin ~2 GB
    behind a KV pool that was sized from free memory); the key holds the torch Stream OBJECT, so its handle cannot be handed to
    another stream while an entry is alive."""
    stream = torch.cuda.current_stream(device)
    key = (device.type, device.index, stream, int(M), int(Cw), int(F))
    sc = _ENC_SCRATCH.pop(key, None)
    if sc is None:
        sc = EncoderScratch(M, Cw, F, device)
    _ENC_SCRATCH[key] = sc                                                  # most recently used last
    while len(_ENC_SCRATCH) > 1 and (len(_ENC_SCRATCH) > 4 or sum(_enc_scratch_bytes(v) for v in _ENC_SCRATCH.values()) > ENC_SCRATCH_CAP_BYTES):
        _ENC_SCRATCH.pop(next(iter(_ENC_SCRATCH)))                          # evict the least recently used; the current one stays
    sc.ws = _scratch(device, min(8 * 4 * M * max(3 * Cw, F), 96 << 20))     # (the shared split-K scratch may have been regrown)
    return sc


def encoder_layer(x, h_in, h_out, L, sc, *, heads, B, act, eps, next_norm=None, p=None, bias_u=None, bias_v=None, klen=-1, chunk=0,
                  left=-1, planes=False):
    """One pre-norm transformer block in ONE library call (vh_encoder_layer): x [M, C] updated in place, h_in = LN(x; norm1),
    h_out = LN(x_out; next_norm) when next_norm = (w, b) is given.  L: dict of this layer's weights (qkv_w, qkv_b, proj_w,
    proj_b, ls1?, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, ls2?).  planes=True: h_in / h_out hold bf16 hi/lo planes (split_planes_into) in
    their M * C * 4 bytes and the Linears run on the weight-streaming GEMM (vita_hip.h: vh_encoder_layer_args.planes)."""
    _dev(x, h_in)
    a = _lib.EncoderLayerArgs()
    a.x, a.h_in, a.h_out = x.data_ptr(), h_in.data_ptr(), (h_out.data_ptr() if next_norm is not None else None)
    a.M, a.C, a.F, a.heads, a.B = sc.M, sc.C, sc.F, int(heads), int(B)
    if tuple(x.shape) != (sc.M, sc.C) or not x.is_contiguous() or not h_in.is_contiguous():
        raise ValueError("encoder_layer: x / h_in must be contiguous [M, C] rows matching the scratch")
    a.qkv_w, a.qkv_b = L["qkv_w"].data_ptr(), L["qkv_b"].data_ptr()
    a.proj_w, a.proj_b, a.ls1 = L["proj_w"].data_ptr(), L["proj_b"].data_ptr(), _p(L.get("ls1"))
    a.n2_w, a.n2_b = L["n2w"].data_ptr(), L["n2b"].data_ptr()
    a.fc1_w, a.fc1_b = L["fc1_w"].data_ptr(), L["fc1_b"].data_ptr()
    a.fc2_w, a.fc2_b, a.ls2 = L["fc2_w"].data_ptr(), L["fc2_b"].data_ptr(), _p(L.get("ls2"))
    if next_norm is not None:
        a.next_w, a.next_b = next_norm[0].data_ptr(), _p(next_norm[1])
    a.act, a.eps = ACT[act], float(eps)
    if p is not None:
        a.P, a.ldp, a.bias_u, a.bias_v = p.data_ptr(), int(p.stride(0)), bias_u.data_ptr(), bias_v.data_ptr()
    a.klen, a.chunk, a.left = int(klen), int(chunk), int(left)
    a.qkv, a.attn, a.hmid, a.mid = sc.qkv.data_ptr(), sc.attn.data_ptr(), sc.hmid.data_ptr(), sc.mid.data_ptr()
    a.ws, a.ws_bytes = sc.ws.data_ptr(), sc.ws.numel() * sc.ws.element_size()
    a.planes = int(bool(planes))
    check(_lib.load().vh_encoder_layer(C.byref(a), _stream()), "vh_encoder_layer")


def layernorm(x, w, b, eps, *, act=None, post_scale=1.0, out=None):
    _dev(x, w)
    _c(x, "x")
    rows, cols = x.shape[0], x.shape[1]
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    check(_lib.load().vh_layernorm(_p(x), x.stride(0), _p(out), out.stride(0), _p(w), _p(b), rows, cols, eps,
                                   ACT[act], post_scale, _stream()), "vh_layernorm")
    return out


def rmsnorm(x, w, eps, out=None):
    _dev(x, w)
    x = _dense(x)
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().vh_rmsnorm(_p(x), _p(out), _p(w), x.shape[0], x.shape[1], eps, _stream()), "vh_rmsnorm")
    return out


def vit_patchify(pix, patch, kpad):
    _dev(pix)
    pix = _dense(pix)
    n, _, img, _ = pix.shape
    g = img // patch
    out = torch.empty((n * g * g, kpad), dtype=torch.float32, device=pix.device)
    check(_lib.load().vh_vit_patchify(_p(pix), _p(out), n, img, patch, kpad, _stream()), "vh_vit_patchify")
    return out


def vit_embed(pix, patch, kpad, patch_w, patch_b, cls, pos, ln_w, ln_b, eps, ntok, x, h, h_planes=None):
    """InternVisionEmbeddings + the first block's norm1 in one library call (vh_vit_embed): pix [n,3,S,S] -> x [n*ntok, C] (residual
    stream), h = LayerNorm(x) fp32 and, when h_planes (any dense buffer of n*ntok*C*4 bytes) is given, its bf16 hi/lo planes."""
    _dev(pix, x, h)
    pix = _dense(pix)
    n, _, img, _ = pix.shape
    g = img // patch
    Cw = int(patch_w.shape[0])
    patches = torch.empty((n * g * g, kpad), dtype=torch.float32, device=pix.device)
    pe = torch.empty((n * g * g, Cw), dtype=torch.float32, device=pix.device)
    M = n * g * g
    ws = _scratch(pix.device, min(8 * 4 * M * Cw, 96 << 20))
    a = _lib.VitEmbedArgs()
    a.pix, a.n, a.img, a.patch, a.kpad = pix.data_ptr(), n, img, int(patch), int(kpad)
    a.patch_w, a.patch_b, a.cls, a.pos = patch_w.data_ptr(), patch_b.data_ptr(), cls.data_ptr(), pos.data_ptr()
    a.ln_w, a.ln_b, a.eps = ln_w.data_ptr(), _p(ln_b), float(eps)
    a.ntok, a.C = int(ntok), Cw
    a.patches, a.pe, a.x, a.h = patches.data_ptr(), pe.data_ptr(), x.data_ptr(), h.data_ptr()
    a.h_planes = h_planes.data_ptr() if h_planes is not None else None
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    check(_lib.load().vh_vit_embed(C.byref(a), _stream()), "vh_vit_embed")
    return x, h


def vit_assemble(patches, cls, pos, n, ntok, hid):
    x = torch.empty((n * ntok, hid), dtype=torch.float32, device=patches.device)
    check(_lib.load().vh_vit_assemble(_p(patches), _p(cls), _p(pos), _p(x), n, ntok, hid, _stream()),
          "vh_vit_assemble")
    return x


def vit_pixel_shuffle(x, n, grid, hid, mul):
    g2 = grid // 2
    out = torch.empty((n, g2 * g2, 4 * hid), dtype=torch.float32, device=x.device)
    check(_lib.load().vh_vit_pixel_shuffle(_p(x), _p(out), n, grid, hid, mul, _stream()), "vh_vit_pixel_shuffle")
    return out


def audio_conv1(feats, mean, istd, w, b):
    _dev(feats, w)
    feats = _dense(feats)
    T, F = feats.shape
    Cc = w.shape[0]
    T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    out = torch.empty((T1 * F1, Cc), dtype=torch.float32, device=feats.device)
    check(_lib.load().vh_audio_conv1(_p(feats), _p(mean), _p(istd), _p(w), _p(b), _p(out), T, F, Cc, _stream()),
          "vh_audio_conv1")
    return out, T1, F1


def embed_splice(kind, idx, embed, img, aud, H):
    S = kind.shape[0]
    out = torch.empty((S, H), dtype=torch.float32, device=embed.device)
    check(_lib.load().vh_embed_splice(_p(kind), _p(idx), _p(embed), _p(img), _p(aud), _p(out), S, H, _stream()),
          "vh_embed_splice")
    return out


def split_planes(x):
    """fp32 [rows, cols] -> (hi, lo) bf16 planes with x = hi + lo to 2^-17 (the pre-split GEMM operand)."""
    _dev(x)
    _f32(_c(x, "x"), "x")
    rows, cols = x.shape
    hi = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    lo = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    check(_lib.load().vh_split_planes(_p(x), x.stride(0), _p(hi), _p(lo), cols, rows, cols, _stream()),
          "vh_split_planes")
    return hi, lo


def split_planes_into(x, buf):
    """fp32 x [rows, cols] -> bf16 hi plane then lo plane inside `buf` (any dense tensor of rows * cols * 4 bytes): the operand layout of
    vh_encoder_layer's planes mode."""
    _dev(x, buf)
    _f32(_c(x, "x"), "x")
    rows, cols = x.shape
    if buf.numel() * buf.element_size() < rows * cols * 4 or not buf.is_contiguous():
        raise ValueError("split_planes_into: buf must be a dense tensor of at least rows * cols * 4 bytes")
    base = buf.data_ptr()
    check(_lib.load().vh_split_planes(_p(x), x.stride(0), base, base + rows * cols * 2, cols, rows, cols, _stream()), "vh_split_planes")
    return buf


def gemm_ps(a_hi, a_lo, w, *, w_up=None, bias=None, act=None, scale=None, resid=None, out=None, out_split=False,
            a_rowidx=None, m=None, c_rowidx=None, group_off=None, ngroups=0, w_group_stride=0, wide=False, ksplit=1,
            nslab_out=None):
    """Weight-streaming GEMM on pre-split activations (vh_gemm_ps).  Returns fp32 out, or (hi, lo) planes when
    out_split=True.  ksplit > 1: `out` must be [ksplit, rows, N] (partial sums per K range; the caller adds them)."""
    _dev(a_hi, a_lo, w)
    _bf16(a_hi, "a_hi"); _bf16(a_lo, "a_lo"); _bf16(w, "w")
    g = _lib.GemmPsArgs()
    N, K = int(w.shape[-2]), int(w.shape[-1])
    g.A_hi, g.A_lo, g.lda = a_hi.data_ptr(), a_lo.data_ptr(), int(a_hi.stride(0))
    g.a_rowidx = a_rowidx.data_ptr() if a_rowidx is not None else None
    g.W = w.data_ptr(); g.W_up = w_up.data_ptr() if w_up is not None else None
    g.ldw, g.w_group_stride = K, int(w_group_stride)
    g.group_off = group_off.data_ptr() if group_off is not None else None
    g.ngroups = int(ngroups)
    M = int(m if m is not None else (a_rowidx.shape[0] if a_rowidx is not None else a_hi.shape[0]))
    planes = None
    if out_split:
        planes = (torch.empty((M, N), dtype=torch.bfloat16, device=w.device),
                  torch.empty((M, N), dtype=torch.bfloat16, device=w.device))
        g.C_hi, g.C_lo, g.ldc_split = planes[0].data_ptr(), planes[1].data_ptr(), N
    else:
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=w.device)
        if ksplit > 1 or ksplit < 0:
            if out.ndim != 3 or out.shape[0] != abs(ksplit):
                raise ValueError("ksplit > 1 needs out of shape [ksplit, rows, N] (ksplit < 0: [-ksplit, rows, N])")
            g.C, g.ldc, g.c_split_stride = out.data_ptr(), int(out.stride(1)), int(out.stride(0))
        else:
            g.C, g.ldc = out.data_ptr(), int(out.stride(0))
    g.ksplit = int(ksplit)
    g.nslab_out = nslab_out.data_ptr() if nslab_out is not None else None
    g.c_rowidx = c_rowidx.data_ptr() if c_rowidx is not None else None
    g.bias = bias.data_ptr() if bias is not None else None
    g.scale = scale.data_ptr() if scale is not None else None
    if resid is not None:
        g.resid, g.ldr = resid.data_ptr(), int(resid.stride(0))
    g.M, g.N, g.K, g.act, g.wide = M, N, K, ACT[act], int(bool(wide))
    check(_lib.load().vh_gemm_ps(C.byref(g), _stream()), "vh_gemm_ps")
    return planes if out_split else out


# ---- batch-1 decode operators, one by one (include/vita_hip.h: "batch-1 decode operators") --------------------------------
def router_top2(x, wg, want_probs=False):
    """MixtralSparseMoeBlock's gate on normed rows x [rows, H] fp32, wg bf16 [E, H] -> (ids int32 [rows, 2], weights fp32
    [rows, 2][, softmax fp32 [rows, E]])."""
    _dev(x, wg)
    _f32(_c(x, "x"), "x"); _bf16(_c(wg, "wg"), "wg")
    rows, H = x.shape
    E = wg.shape[0]
    ids = torch.empty((rows, 2), dtype=torch.int32, device=x.device)
    wts = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    probs = torch.empty((rows, E), dtype=torch.float32, device=x.device) if want_probs else None
    check(_lib.load().vh_router_top2(_p(x), x.stride(0), _p(wg), E, H, rows, _p(ids), _p(wts), _p(probs), _stream()), "vh_router_top2")
    return (ids, wts, probs) if want_probs else (ids, wts)


def moe_decode(x, norm_w, eps, wg, w1, w3, w2, delta=None):
    """block_sparse_moe(post_attention_layernorm(x + delta)) for ONE token (no residual add): x fp32 [H], w1 / w3 bf16
    [E, I, H], w2 bf16 [E, H, I] -> (y fp32 [H], route int32 [4] = e0, e1, bits(w0), bits(w1))."""
    _dev(x, wg, w1, w3, w2)
    _f32(_c(x, "x"), "x")
    E, I, H = w1.shape
    y = torch.empty(H, dtype=torch.float32, device=x.device)
    route = torch.zeros(4, dtype=torch.int32, device=x.device)
    hbuf = torch.empty(2 * I, dtype=torch.float32, device=x.device)
    check(_lib.load().vh_moe_decode(_p(x), _p(delta), _p(norm_w), float(eps), _p(wg), _p(w1), _p(w3), _p(w2), E, I, H, None, _p(y),
                                    _p(route), _p(hbuf), _stream()), "vh_moe_decode")
    return y, route


def rope_kv_append(qkv, kcache, vcache, rope_cos, rope_sin, pos0, nq, nkv, table=None):
    """RoPE on q, k of S fused-QKV rows [S, (nq + 2 nkv) * 128] + KV append at [pos0, pos0 + S) -> q_roped fp32 [S, nq * 128];
    kcache / vcache fp32 [nkv, max_ctx, 128] are updated in place."""
    _dev(qkv, kcache, vcache)
    _f32(_c(qkv, "qkv"), "qkv")
    Sn = qkv.shape[0]
    q = torch.empty((Sn, nq * 128), dtype=torch.float32, device=qkv.device)
    check(_lib.load().vh_rope_kv_append(_p(qkv), qkv.stride(0), _p(q), _p(kcache), _p(vcache), _p(rope_cos), _p(rope_sin), Sn, int(pos0),
                                        nq, nkv, kcache.shape[1], _p(table), _stream()), "vh_rope_kv_append")
    return q


def attn_decode(qkv, kcache, vcache, pos, rope_cos, rope_sin, nq, nkv, scale, table=None):
    """one token's RoPE + KV append at `pos` + causal GQA attention over [0, pos] -> fp32 [nq * 128]."""
    _dev(qkv, kcache, vcache)
    max_ctx = kcache.shape[1]
    ns = (max_ctx + 63) // 64
    dev = qkv.device
    part_o = torch.empty((nq, ns, 128), dtype=torch.float32, device=dev)
    part_ml = torch.empty((nq, ns, 2), dtype=torch.float32, device=dev)
    tickets = torch.zeros(nkv, dtype=torch.int32, device=dev)
    out = torch.empty(nq * 128, dtype=torch.float32, device=dev)
    check(_lib.load().vh_attn_decode(_p(qkv), _p(kcache), _p(vcache), int(pos), _p(rope_cos), _p(rope_sin), nq, nkv, max_ctx, float(scale),
                                     _p(table), _p(part_o), _p(part_ml), _p(tickets), _p(out), _stream()), "vh_attn_decode")
    return out


def lmhead_argmax(x, norm_w, eps, w, delta=None, nblk=1024):
    """logits = lm_head(rmsnorm(x + delta)) fp32 [V] and their argmax (int32 [1], lowest index on ties)."""
    _dev(x, w)
    V, H = w.shape
    logits = torch.empty(V, dtype=torch.float32, device=x.device)
    tok = torch.zeros(1, dtype=torch.int32, device=x.device)
    bv = torch.empty(nblk, dtype=torch.float32, device=x.device)
    bi = torch.empty(nblk, dtype=torch.int32, device=x.device)
    check(_lib.load().vh_lmhead_argmax(_p(x), _p(delta), _p(norm_w), float(eps), _p(w), V, H, _p(logits), _p(tok), _p(bv), _p(bi), nblk,
                                       _stream()), "vh_lmhead_argmax")
    return logits, tok
