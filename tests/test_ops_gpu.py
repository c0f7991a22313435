"""GPU parity of every generic C-ABI operator against plain numpy (the per-kernel oracle):
fp32 references on the same bf16-rounded weights.  Tolerances: the split-bf16 GEMM carries the
activation to 2^-17, so |err| <= ~2e-5 * sum|a*w| in the worst case; stated per test."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, to_np
from vita_amd.checkpoint import round_bf16

pytestmark = pytest.mark.gpu

# plain / causal attention runs on bf16 x 3 MFMAs (vh_attn.hip k_attn_x3): every q.k and p.v product is exact to ~2^-17 of its
# magnitude, as in the exact-mode GEMMs; at N(0,1) operands and d <= 128 that is < 1e-4 on the output
ATTN_X3_ATOL = 1e-4


def _w(rng, *shape, std=0.05):
    return round_bf16(rng.standard_normal(shape, dtype=np.float32) * std)


def _dev(x, dev, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev).to(dtype)


def gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


ACTS = {None: lambda x: x, "gelu": gelu, "relu": lambda x: np.maximum(x, 0), "silu": lambda x: x / (1 + np.exp(-x))}


# ---------------------------------------------------------------------------------------------
def test_mfma_layout_probe(dev):
    """A = I-like pattern with an ASYMMETRIC W catches row/col swaps of the MFMA fragment maps."""
    from vita_amd import ops
    M, N, K = 64, 128, 64
    a = np.zeros((M, K), np.float32)
    a[np.arange(M), np.arange(M) % K] = 1.0
    w = round_bf16((np.arange(N)[:, None] * 0.5 + np.arange(K)[None, :] * 0.001953125).astype(np.float32))
    got = to_np(ops.gemm(_dev(a, dev), _dev(w, dev, torch.bfloat16)))
    assert_close("mfma_layout", got, a.astype(np.float64) @ w.T.astype(np.float64), atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(1, 64, 64), (7, 100, 128), (64, 128, 64), (130, 257, 192), (450, 6144, 4096),
                                   (1025, 3072, 1024)])
def test_gemm_plain(dev, M, N, K):
    from vita_amd import ops
    rng = np.random.default_rng(M * 7 + N)
    a = rng.standard_normal((M, K), dtype=np.float32)
    w = _w(rng, N, K)
    got = to_np(ops.gemm(_dev(a, dev), _dev(w, dev, torch.bfloat16)))
    ref = a.astype(np.float64) @ w.T.astype(np.float64)
    assert_close(f"gemm {M}x{N}x{K}", got, ref, atol=2e-5 * np.sqrt(K), rtol=2e-5)


@pytest.mark.parametrize("act", [None, "gelu", "relu", "silu"])
def test_gemm_epilogue(dev, act):
    from vita_amd import ops
    rng = np.random.default_rng(3)
    M, N, K = 77, 200, 128
    a = rng.standard_normal((M, K), dtype=np.float32)
    w = _w(rng, N, K)
    bias, scale = _w(rng, N, std=0.5), _w(rng, N, std=1.0)
    resid = rng.standard_normal((M, N), dtype=np.float32)
    got = to_np(ops.gemm(_dev(a, dev), _dev(w, dev, torch.bfloat16), bias=_dev(bias, dev), act=act,
                         scale=_dev(scale, dev), resid=_dev(resid, dev)))
    ref = ACTS[act](a.astype(np.float64) @ w.T + bias) * scale + resid
    assert_close(f"gemm epilogue {act}", got, ref, atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("ksplit", [0, 1, 3, 8])
def test_gemm_split_k(dev, ksplit):
    """split-K (partial slabs + reducing kernel with the whole epilogue) against the fp64 reference, on the encoder shape
    that leaves half the chip idle unsplit (M = 1025, N = 1024, K = 4096: 136 blocks), in-place residual included."""
    from vita_amd import ops
    rng = np.random.default_rng(40)
    M, N, K = 1025, 1024, 4096
    a = rng.standard_normal((M, K), dtype=np.float32)
    w = _w(rng, N, K)
    bias, scale = _w(rng, N, std=0.5), _w(rng, N, std=1.0)
    x = rng.standard_normal((M, N), dtype=np.float32)
    xd = _dev(x, dev)
    ops.gemm(_dev(a, dev), _dev(w, dev, torch.bfloat16), bias=_dev(bias, dev), act="gelu", scale=_dev(scale, dev),
             resid=xd, out=xd, ksplit=ksplit)
    ref = ACTS["gelu"](a.astype(np.float64) @ w.T + bias) * scale + x
    assert_close(f"gemm split-K {ksplit}", to_np(xd), ref, atol=2e-5 * np.sqrt(K), rtol=2e-5)
    # a small-K launch never splits below 4 K-tiles per block, a ragged N (not % 4) never splits
    a2 = rng.standard_normal((70, 192), dtype=np.float32)
    w2 = _w(rng, 130, 192)
    got = to_np(ops.gemm(_dev(a2, dev), _dev(w2, dev, torch.bfloat16), ksplit=ksplit))
    assert_close("ragged", got, a2.astype(np.float64) @ w2.T.astype(np.float64), atol=1e-4, rtol=2e-5)


@pytest.mark.parametrize("M,N,K,ksplit,with_bias", [(1025, 1024, 1024, 0, True), (249, 1024, 4096, 0, True), (249, 1024, 1024, 1, False),
                                                   (5125, 1024, 1024, 0, True), (33, 256, 128, 0, False)])
def test_gemm_with_layernorm(dev, M, N, K, ksplit, with_bias):
    """vh_gemm_ln: the Linear (+ bias, layer scale, in-place residual) and the LayerNorm of its output in one call — through
    the split-K reducer that norms whole rows (one tile / one clip), and through the norm launch that follows an unsplit
    GEMM (5 tiles; ksplit = 1; tiny K) — against the fp64 composition (modeling_intern_vit.py:245-253)."""
    from vita_amd import ops
    rng = np.random.default_rng(M + N)
    a = rng.standard_normal((M, K), dtype=np.float32)
    w = _w(rng, N, K)
    bias, scale = _w(rng, N, std=0.5), _w(rng, N, std=1.0)
    lw, lb = 1.0 + _w(rng, N, std=0.2), _w(rng, N, std=0.3)
    x = rng.standard_normal((M, N), dtype=np.float32)
    xd = _dev(x, dev)
    out, h = ops.gemm(_dev(a, dev), _dev(w, dev, torch.bfloat16), bias=_dev(bias, dev), scale=_dev(scale, dev), resid=xd, out=xd,
                      ksplit=ksplit, ln=(_dev(lw, dev), _dev(lb, dev) if with_bias else None, 1e-6))
    assert out.data_ptr() == xd.data_ptr()
    ref = (a.astype(np.float64) @ w.T + bias) * scale + x
    mu, var = ref.mean(-1, keepdims=True), ref.var(-1, keepdims=True)
    ref_h = (ref - mu) / np.sqrt(var + 1e-6) * lw + (lb if with_bias else 0.0)
    assert_close("gemm_ln: C", to_np(xd), ref, atol=2e-5 * np.sqrt(K), rtol=2e-5)
    assert_close("gemm_ln: LayerNorm(C)", to_np(h), ref_h, atol=5e-5 * np.sqrt(K), rtol=2e-5)


def test_gemm_inplace_residual(dev):
    """out aliases resid (the x += proj(...) pattern)."""
    from vita_amd import ops
    rng = np.random.default_rng(4)
    M, N, K = 90, 128, 64
    a = rng.standard_normal((M, K), dtype=np.float32)
    w = _w(rng, N, K)
    x = rng.standard_normal((M, N), dtype=np.float32)
    xd = _dev(x, dev)
    ops.gemm(_dev(a, dev), _dev(w, dev, torch.bfloat16), resid=xd, out=xd)
    assert_close("gemm inplace resid", to_np(xd), x + a.astype(np.float64) @ w.T, atol=1e-4)


def test_gemm_segments_conv(dev):
    """conv2d 3x3 stride 2 over a channels-last map expressed as row-offset segments
    (whale subsampling conv2: subsampling.py:28-43)."""
    from vita_amd import ops
    rng = np.random.default_rng(5)
    T1, F1, Cin, Cout = 13, 9, 64, 96
    x = rng.standard_normal((T1, F1, Cin), dtype=np.float32)
    wt = _w(rng, Cout, Cin, 3, 3)  # torch layout [Cout, Cin, kh, kw]
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    ref = np.zeros((T2, F2, Cout))
    for t in range(T2):
        for f in range(F2):
            patch = x[2 * t:2 * t + 3, 2 * f:2 * f + 3, :]  # [kh, kw, Cin]
            ref[t, f] = np.einsum("hwc,ochw->o", patch.astype(np.float64), wt.astype(np.float64))
    wp = np.ascontiguousarray(wt.transpose(0, 2, 3, 1).reshape(Cout, 9 * Cin))  # [Cout, (kh,kw,cin)]
    rowidx = np.array([(2 * t) * F1 + 2 * f for t in range(T2) for f in range(F2)], np.int32)
    segrow = [kh * F1 + kw for kh in range(3) for kw in range(3)]
    got = to_np(ops.gemm(_dev(x.reshape(T1 * F1, Cin), dev), _dev(wp, dev, torch.bfloat16),
                         a_rowidx=_dev(rowidx, dev, torch.int32), segrow=segrow, seglen=Cin))
    assert_close("gemm conv segments", got, ref.reshape(T2 * F2, Cout), atol=1e-4)


def test_gemm_segments_zero_pad(dev):
    """conv1d k=5 s=2 with right zero padding via out-of-range rows (whale adapter.py:107-127)."""
    from vita_amd import ops
    rng = np.random.default_rng(6)
    T, Cin, Cout, k = 11, 64, 80, 5
    x = rng.standard_normal((T, Cin), dtype=np.float32)
    wt = _w(rng, Cout, Cin, k)
    xp = np.concatenate([x, np.zeros((k - 1, Cin), np.float32)])
    To = (T + k - 1 - k) // 2 + 1
    ref = np.stack([np.einsum("kc,ock->o", xp[2 * t:2 * t + k].astype(np.float64), wt.astype(np.float64))
                    for t in range(To)])
    wp = np.ascontiguousarray(wt.transpose(0, 2, 1).reshape(Cout, k * Cin))
    rowidx = (2 * np.arange(To)).astype(np.int32)
    got = to_np(ops.gemm(_dev(x, dev), _dev(wp, dev, torch.bfloat16), a_rowidx=_dev(rowidx, dev, torch.int32),
                         segrow=list(range(k)), seglen=Cin))  # rows >= T read as zero
    assert_close("gemm conv1d zero pad", got, ref, atol=1e-4)


def test_gemm_grouped_glu_and_scatter(dev):
    """The top-2 MoE pair: grouped gate/up GEMM with SiLU*up, then grouped down GEMM scattered to
    (token, slot) rows — HF MixtralExperts (modeling_mixtral.py:57-93)."""
    from vita_amd import ops
    rng = np.random.default_rng(7)
    S, H, I, E = 150, 128, 192, 4
    x = rng.standard_normal((S, H), dtype=np.float32)
    w1, w3, w2 = _w(rng, E, I, H), _w(rng, E, I, H), _w(rng, E, H, I)
    ids = np.stack([rng.permutation(E)[:2] for _ in range(S)]).astype(np.int32)
    ids[:10] = [[2, 0]] * 10  # skew: expert 1 may get few rows
    flat = ids.reshape(-1)
    order = np.argsort(flat, kind="stable")
    goff = np.concatenate([[0], np.cumsum(np.bincount(flat, minlength=E))]).astype(np.int32)
    stok, sslot = (order // 2).astype(np.int32), order.astype(np.int32)
    h = ops.gemm(_dev(x, dev), _dev(w1, dev, torch.bfloat16), w_up=_dev(w3, dev, torch.bfloat16),
                 a_rowidx=_dev(stok, dev, torch.int32), group_off=_dev(goff, dev, torch.int32), ngroups=E,
                 w_group_stride=I * H, m=2 * S)
    href = np.zeros((2 * S, I))
    for p, slot in enumerate(order):
        e, t = flat[slot], slot // 2
        g, u = x[t].astype(np.float64) @ w1[e].T, x[t].astype(np.float64) @ w3[e].T
        href[p] = g / (1 + np.exp(-g)) * u
    assert_close("grouped GLU", to_np(h), href, atol=2e-4)
    y = torch.zeros((2 * S, H), dtype=torch.float32, device=dev)
    ops.gemm(h, _dev(w2, dev, torch.bfloat16), group_off=_dev(goff, dev, torch.int32), ngroups=E,
             w_group_stride=H * I, c_rowidx=_dev(sslot, dev, torch.int32), out=y)
    yref = np.zeros((2 * S, H))
    hn = to_np(h).astype(np.float64)
    for p, slot in enumerate(order):
        yref[slot] = hn[p] @ w2[flat[slot]].T
    assert_close("grouped down + scatter", to_np(y), yref, atol=2e-4)


@pytest.fixture(params=[-1, 0, 1, 2], ids=["cfg-auto", "cfg-dma-ring", "cfg-regstaged", "cfg-specialised"])
def ps_cfg(request):
    """every vh_gemm_ps test runs on the default selection (by rows per group) and on each kernel form forced: 64-row tiles
    with the weights in an LDS-DMA ring, 192-row tiles with register-staged weights (all 8 waves load and multiply), and the
    12-wave form with dedicated loader waves (vh_gemm_sp.hip)."""
    from vita_amd import _lib
    _lib.tune("ps_cfg", request.param)
    yield request.param
    _lib.tune("ps_cfg", -1)


def test_gemm_ps_ksplit_slabs(dev, ps_cfg):
    """K-split down projection: partial slabs per K range add up to the unsplit product; ragged expert sizes,
    an empty expert, more experts than XCD runs."""
    from vita_amd import ops
    rng = np.random.default_rng(21)
    M, H, I, E = 200, 192, 448, 6
    h = rng.standard_normal((M, I), dtype=np.float32)
    w2 = _w(rng, E, H, I)
    cnt = np.array([70, 0, 33, 1, 80, 16])
    goff = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    perm = rng.permutation(M).astype(np.int32)
    hh, hl = ops.split_planes(_dev(h, dev))
    for ks in (1, 2, 3):
        y = torch.full((ks, M, H), 7.0, dtype=torch.float32, device=dev)
        ops.gemm_ps(hh, hl, _dev(w2, dev, torch.bfloat16), group_off=_dev(goff, dev, torch.int32), ngroups=E,
                    w_group_stride=H * I, c_rowidx=_dev(perm, dev, torch.int32), out=y if ks > 1 else y[0], ksplit=ks)
        ref = np.zeros((M, H))
        for e in range(E):
            for p in range(goff[e], goff[e + 1]):
                ref[perm[p]] = h[p].astype(np.float64) @ w2[e].T
        assert_close(f"ksplit {ks}", to_np(y.sum(0)), ref, atol=2e-4)


@pytest.mark.parametrize("cnt", [[2, 0, 1, 0, 0, 2, 0, 1], [1, 1, 1, 0, 1, 1, 0, 1], [3, 2, 4, 1, 2, 1, 2, 1], [6, 0, 0, 0, 0, 0, 0, 0]])
def test_gemm_ps_device_chosen_ksplit_few_rows(dev, cnt):
    """the down projection of an iteration of a few concurrent sequences (vh_api.hip: decode_iteration): every expert holds a handful of rows,
    the kernel picks the K split itself (ksplit = -4) and reports it; the slabs it wrote add up to the unsplit product.  With one row tile per
    expert a partial round cannot be M-split, so 4 touched experts x 16 n-tiles take 4 slabs (256 tiles: the whole chip), not 2."""
    from vita_amd import ops
    rng = np.random.default_rng(sum(cnt) * 31 + cnt[0])
    E, H, I = 8, 4096, 1024
    M = int(sum(cnt))
    h = rng.standard_normal((M, I), dtype=np.float32)
    w2 = _w(rng, E, H, I)
    goff = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    perm = rng.permutation(M).astype(np.int32)
    hh, hl = ops.split_planes(_dev(h, dev))
    y = torch.full((4, M, H), 7.0, dtype=torch.float32, device=dev)
    nslab = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.gemm_ps(hh, hl, _dev(w2, dev, torch.bfloat16), group_off=_dev(goff, dev, torch.int32), ngroups=E, w_group_stride=H * I,
                c_rowidx=_dev(perm, dev, torch.int32), out=y, ksplit=-4, nslab_out=nslab)
    ks = int(nslab.item())
    assert 1 <= ks <= 4
    ref = np.zeros((M, H))
    for e in range(E):
        for p in range(goff[e], goff[e + 1]):
            ref[perm[p]] = h[p].astype(np.float64) @ w2[e].T
    assert_close(f"device-chosen ksplit {ks}", to_np(y[:ks].sum(0)), ref, atol=3e-4)
    assert torch.all(y[ks:] == 7.0)                                    # slabs above the reported count are not touched
    if torch.cuda.get_device_properties(0).multi_processor_count == 256 and sum(c > 0 for c in cnt) == 4:
        assert ks == 4


def _planes_to_f32(hi, lo):
    return hi.float() + lo.float()


def test_split_planes(dev):
    from vita_amd import ops
    rng = np.random.default_rng(17)
    x = (rng.standard_normal((37, 256), dtype=np.float32) * np.exp(rng.standard_normal((37, 256)) * 3)).astype(np.float32)
    hi, lo = ops.split_planes(_dev(x, dev))
    rec = to_np(_planes_to_f32(hi, lo))
    assert np.all(np.abs(rec - x) <= np.abs(x) * 2.0 ** -16)          # two bf16 terms carry >= 16 mantissa bits
    assert np.array_equal(to_np(hi.float()), round_bf16(x))


@pytest.mark.parametrize("M,N,K,wide", [(1, 64, 64, False), (16, 128, 64, False), (138, 300, 192, False),
                                        (193, 257, 128, True), (400, 512, 4096, True), (552, 4096, 1024, False),
                                        (288, 256, 64, False), (289, 96, 128, False), (250, 512, 320, False), (33, 40, 2048, False)])
def test_gemm_ps_plain(dev, M, N, K, wide, ps_cfg):
    """pre-split skinny GEMM vs fp64 reference on the same bf16 weights: ragged M (row tiles + clamped
    rows), N tails, several 192-row m-tiles, K long enough to wrap the LDS ring many times."""
    from vita_amd import ops
    rng = np.random.default_rng(M * 7 + N)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = _w(rng, N, K)
    b = _w(rng, N)
    r = rng.standard_normal((M, N), dtype=np.float32)
    hi, lo = ops.split_planes(_dev(x, dev))
    y = ops.gemm_ps(hi, lo, _dev(w, dev, torch.bfloat16), bias=_dev(b, dev), act="gelu", resid=_dev(r, dev), wide=wide)
    ref = gelu(x.astype(np.float64) @ w.T.astype(np.float64) + b) + r
    assert_close(f"gemm_ps {M}x{N}x{K}", to_np(y), ref, atol=3e-4 if K > 1024 else 1e-4)


def test_gemm_ps_grouped_glu_and_scatter(dev, ps_cfg):
    """the MoE pair on the pre-split kernel: gather by sorted token, grouped GLU emitting split planes,
    grouped down GEMM scattered to (token, slot) rows; one expert empty, one above 192 rows."""
    from vita_amd import ops
    rng = np.random.default_rng(8)
    S, H, I, E = 330, 128, 256, 5
    x = rng.standard_normal((S, H), dtype=np.float32)
    w1, w3, w2 = _w(rng, E, I, H), _w(rng, E, I, H), _w(rng, E, H, I)
    ids = np.stack([rng.permutation(4)[:2] for _ in range(S)]).astype(np.int32)   # expert 4 never chosen
    ids[:250] = [[1, 3]] * 250                                                      # expert 1 and 3 exceed one m-tile
    flat = ids.reshape(-1)
    order = np.argsort(flat, kind="stable")
    goff = np.concatenate([[0], np.cumsum(np.bincount(flat, minlength=E))]).astype(np.int32)
    stok, sslot = (order // 2).astype(np.int32), order.astype(np.int32)
    xh, xl = ops.split_planes(_dev(x, dev))
    hh, hl = ops.gemm_ps(xh, xl, _dev(w1, dev, torch.bfloat16), w_up=_dev(w3, dev, torch.bfloat16),
                         a_rowidx=_dev(stok, dev, torch.int32), group_off=_dev(goff, dev, torch.int32), ngroups=E,
                         w_group_stride=I * H, m=2 * S, out_split=True)
    href = np.zeros((2 * S, I))
    for p, slot in enumerate(order):
        e, t = flat[slot], slot // 2
        g, u = x[t].astype(np.float64) @ w1[e].T, x[t].astype(np.float64) @ w3[e].T
        href[p] = g / (1 + np.exp(-g)) * u
    h = _planes_to_f32(hh, hl)
    assert_close("ps grouped GLU (planes)", to_np(h), href, atol=2e-4)
    y = torch.zeros((2 * S, H), dtype=torch.float32, device=dev)
    ops.gemm_ps(hh, hl, _dev(w2, dev, torch.bfloat16), group_off=_dev(goff, dev, torch.int32), ngroups=E,
                w_group_stride=H * I, c_rowidx=_dev(sslot, dev, torch.int32), out=y)
    yref = np.zeros((2 * S, H))
    hn = to_np(h).astype(np.float64)
    for p, slot in enumerate(order):
        yref[slot] = hn[p] @ w2[flat[slot]].T
    assert_close("ps grouped down + scatter", to_np(y), yref, atol=2e-4)


# ---------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, scale, mask=None, p=None, bu=None, bv=None):
    """q [H,Sq,d], k/v [Hkv,Sk,d]; mask [Sq,Sk] True = visible."""
    H, Sq, d = q.shape
    g = H // k.shape[0]
    out = np.zeros((Sq, H * d))
    for h in range(H):
        if p is None:
            s = q[h].astype(np.float64) @ k[h // g].T
        else:
            s = (q[h] + bu[h]).astype(np.float64) @ k[h // g].T + (q[h] + bv[h]).astype(np.float64) @ p[h].T
        s = s * scale
        if mask is not None:
            s = np.where(mask, s, -np.inf)
        m = np.max(s, axis=-1, keepdims=True)
        m = np.where(np.isfinite(m), m, 0.0)
        e = np.exp(s - m)
        den = e.sum(-1, keepdims=True)
        pr = np.where(den > 0, e / np.where(den > 0, den, 1), 0.0)
        out[:, h * d:(h + 1) * d] = pr @ v[h // g]
    return out


@pytest.mark.parametrize("rows,N,groups", [(16, 77, 0), (32, 77, 0), (32, 333, 4), (32, 1025, 0), (32, 97, 2)])
def test_attention_vit_like(dev, rows, N, groups):
    """non-causal, 2 images x 3 heads x 64, N=77 (tails in both q and key tiles), packed qkv rows; 16 and 32 query rows per wave
    (attn_rows), the ViT's N = 1025 (one valid row in the last 32-row block), key groups merged across two row tiles."""
    from vita_amd import _lib, ops
    rng = np.random.default_rng(8)
    B, H, d = 2, 3, 64
    _lib.tune("attn_rows", rows)
    _lib.tune("attn_ksplit", groups)
    try:
        _attention_vit_like(dev, rng, B, H, N, d)
    finally:
        _lib.tune("attn_rows", 0)
        _lib.tune("attn_ksplit", 0)


def _attention_vit_like(dev, rng, B, H, N, d):
    from vita_amd import ops
    qkv = rng.standard_normal((B * N, 3 * H * d), dtype=np.float32)
    t = _dev(qkv, dev)
    out = torch.empty((B * N, H * d), dtype=torch.float32, device=dev)
    ops.attention(t, t[:, H * d:], t[:, 2 * H * d:], out, B=B, Hq=H, Hkv=H, Sq=N, Sk=N, d=d, ldq=3 * H * d, hsq=d,
                  ldk=3 * H * d, hsk=d, ldv=3 * H * d, hsv=d, ldo=H * d, bsq=N * 3 * H * d, bsk=N * 3 * H * d,
                  bso=N * H * d, scale=d ** -0.5)
    for b in range(B):
        r = qkv[b * N:(b + 1) * N].reshape(N, 3, H, d).transpose(1, 2, 0, 3)
        ref = _attn_ref(r[0], r[1], r[2], d ** -0.5)
        assert_close(f"attn vit b{b}", to_np(out[b * N:(b + 1) * N]), ref, atol=ATTN_X3_ATOL)


@pytest.mark.parametrize("impl", [0, 2])
@pytest.mark.parametrize("groups", [1, 2, 4])
def test_attention_key_groups(dev, groups, impl):
    """every key-group instantiation (attn_ksplit: tiles dealt to 1 / 2 / 4 wave groups, merged in group order) gives the
    single-pass result: plain d=64, rel-pos d=64, causal d=128 (4 falls back to 2 there).  impl 0 = the default kernels (plain and
    causal on bf16 x 3 MFMAs: products exact to 2^-17 per term like the GEMMs, tolerance ATTN_X3_ATOL; rel-pos on the fp32 MFMA),
    2 = fp32 MFMA everywhere (fp32 rounding: 2e-5)."""
    from vita_amd import _lib, ops
    rng = np.random.default_rng(80)
    _lib.tune("attn_ksplit", groups)
    _lib.tune("attn_impl", impl)
    tol = ATTN_X3_ATOL if impl == 0 else 2e-5
    try:
        H, N, d = 2, 333, 64
        q, k, v, p = (rng.standard_normal((N, H * d), dtype=np.float32) for _ in range(4))
        bu, bv = rng.standard_normal((H, d), dtype=np.float32), rng.standard_normal((H, d), dtype=np.float32)
        sp = lambda x: x.reshape(N, H, d).transpose(1, 0, 2)
        out = torch.empty((N, H * d), dtype=torch.float32, device=dev)
        kw = dict(B=1, Hq=H, Hkv=H, Sq=N, Sk=N, d=d, ldq=H * d, hsq=d, ldk=H * d, hsk=d, ldv=H * d, hsv=d, ldo=H * d,
                  scale=d ** -0.5)
        ops.attention(_dev(q, dev), _dev(k, dev), _dev(v, dev), out, **kw)
        assert_close(f"plain groups={groups}", to_np(out), _attn_ref(sp(q), sp(k), sp(v), d ** -0.5), atol=tol)
        ops.attention(_dev(q, dev), _dev(k, dev), _dev(v, dev), out, klen=301, p=_dev(p, dev), ldp=H * d, hsp=d,
                      bias_u=_dev(bu, dev), bias_v=_dev(bv, dev), **kw)
        mask = np.broadcast_to(np.arange(N)[None, :] < 301, (N, N))
        ref = _attn_ref(sp(q), sp(k), sp(v), d ** -0.5, mask, sp(p), bu[:, None, :], bv[:, None, :])
        assert_close(f"relpos groups={groups}", to_np(out), ref, atol=5e-5)
        nq, nkv, d2, max_ctx, Sq, pos0 = 4, 2, 128, 400, 150, 170
        Sk = pos0 + Sq
        q2 = rng.standard_normal((Sq, nq * d2), dtype=np.float32)
        kc = rng.standard_normal((nkv, max_ctx, d2), dtype=np.float32)
        vc = rng.standard_normal((nkv, max_ctx, d2), dtype=np.float32)
        out2 = torch.empty((Sq, nq * d2), dtype=torch.float32, device=dev)
        ops.attention(_dev(q2, dev), _dev(kc, dev), _dev(vc, dev), out2, B=1, Hq=nq, Hkv=nkv, Sq=Sq, Sk=Sk, d=d2,
                      ldq=nq * d2, hsq=d2, ldk=d2, hsk=max_ctx * d2, ldv=d2, hsv=max_ctx * d2, ldo=nq * d2,
                      scale=d2 ** -0.5, causal=True, q_off=pos0)
        cm = np.arange(Sk)[None, :] <= (pos0 + np.arange(Sq))[:, None]
        ref2 = _attn_ref(q2.reshape(Sq, nq, d2).transpose(1, 0, 2), kc[:, :Sk], vc[:, :Sk], d2 ** -0.5, cm)
        assert_close(f"causal gqa groups={groups}", to_np(out2), ref2, atol=tol)
    finally:
        _lib.tune("attn_ksplit", 0)
        _lib.tune("attn_impl", 0)


@pytest.mark.parametrize("nq,nkv,causal", [(8, 2, True), (4, 1, True), (8, 2, False)])
@pytest.mark.parametrize("Sq,pos0", [(70, 0), (33, 45), (1, 99), (150, 170), (64, 64), (129, 0)])
@pytest.mark.parametrize("rows,xcd", [(0, 1), (32, 1), (32, 0)], ids=["rows16", "rows32", "rows32-plain-grid"])
def test_attention_flash_form(dev, nq, nkv, causal, Sq, pos0, rows, xcd):
    """k_attn_fa forced (attn_fa = 2; by default it runs when its blocks fill half the chip): d = 128, 4 : 1 head grouping (one block
    = the four query heads of a KV head on the same 16 rows, eight waves = heads x two 32-key halves merged at the end), causal with
    a query offset (chunked prefill) or a key pad mask, K / V in cache layout [nkv][max_ctx][d]; tails in the 16-row blocks, the
    64-key tiles and their halves; rows past the context hold NaNs and must never reach a product.  r06: 32 query rows per wave
    (attn_rows = 32, causal only: what long prompts take by themselves) and the XCD-aware block -> (q tile, KV head) mapping on / off."""
    from vita_amd import _lib, ops
    rng = np.random.default_rng(10 + Sq + nq)
    d, max_ctx = 128, 400
    Sk = pos0 + Sq
    klen = Sk if causal else max(1, Sk - 37)
    q = rng.standard_normal((Sq, nq * d), dtype=np.float32)
    kc = rng.standard_normal((nkv, max_ctx, d), dtype=np.float32)
    vc = rng.standard_normal((nkv, max_ctx, d), dtype=np.float32)
    kc[:, Sk:] = np.nan
    vc[:, Sk:] = np.nan
    if not causal:          # pad mask: rows in [klen, Sk) are caller memory too — masked keys must not be loaded from them
        kc[:, klen:] = np.nan
        vc[:, klen:] = np.nan
    out = torch.full((Sq, nq * d), float("nan"), dtype=torch.float32, device=dev)
    _lib.tune("attn_fa", 2)
    _lib.tune("attn_rows", rows)
    _lib.tune("attn_xcd", xcd)
    try:
        ops.attention(_dev(q, dev), _dev(kc, dev), _dev(vc, dev), out, B=1, Hq=nq, Hkv=nkv, Sq=Sq, Sk=Sk, d=d, ldq=nq * d,
                      hsq=d, ldk=d, hsk=max_ctx * d, ldv=d, hsv=max_ctx * d, ldo=nq * d, scale=d ** -0.5, causal=causal, q_off=pos0,
                      klen=klen)
    finally:
        _lib.tune("attn_fa", 1)
        _lib.tune("attn_rows", 0)
        _lib.tune("attn_xcd", 1)
    mask = (np.arange(Sk)[None, :] <= (pos0 + np.arange(Sq))[:, None]) if causal else np.broadcast_to(np.arange(Sk)[None, :] < klen, (Sq, Sk))
    ref = _attn_ref(q.reshape(Sq, nq, d).transpose(1, 0, 2), np.nan_to_num(kc[:, :Sk]), np.nan_to_num(vc[:, :Sk]), d ** -0.5, mask)
    assert_close(f"flash form nq={nq} causal={causal} Sq={Sq} pos0={pos0}", to_np(out), ref, atol=ATTN_X3_ATOL)


def test_attention_big_scores(dev):
    """large score range exercises the online-softmax rescale branch (running max jumps)."""
    from vita_amd import ops
    rng = np.random.default_rng(9)
    H, N, d = 2, 130, 64
    q = rng.standard_normal((N, H * d), dtype=np.float32) * 3
    k = rng.standard_normal((N, H * d), dtype=np.float32) * 3
    k[97] *= 4.0  # spike late in the key order
    v = rng.standard_normal((N, H * d), dtype=np.float32)
    out = torch.empty((N, H * d), dtype=torch.float32, device=dev)
    ops.attention(_dev(q, dev), _dev(k, dev), _dev(v, dev), out, B=1, Hq=H, Hkv=H, Sq=N, Sk=N, d=d, ldq=H * d, hsq=d,
                  ldk=H * d, hsk=d, ldv=H * d, hsv=d, ldo=H * d, scale=d ** -0.5)
    sp = lambda x: x.reshape(N, H, d).transpose(1, 0, 2)
    assert_close("attn big scores", to_np(out), _attn_ref(sp(q), sp(k), sp(v), d ** -0.5), atol=4 * ATTN_X3_ATOL)   # scores x 9


@pytest.mark.parametrize("Sq,pos0", [(70, 0), (33, 45), (1, 99)])
def test_attention_causal_gqa(dev, Sq, pos0):
    """Mixtral prefill shape on the direct kernel: 4 q-heads / 2 kv-heads x 128 (a 2 : 1 grouping the flash form does not take),
    K/V in cache layout [nkv][max_ctx][128]."""
    from vita_amd import _lib, ops
    rng = np.random.default_rng(10 + Sq)
    nq, nkv, d, max_ctx = 4, 2, 128, 160
    Sk = pos0 + Sq
    q = rng.standard_normal((Sq, nq * d), dtype=np.float32)
    kc = rng.standard_normal((nkv, max_ctx, d), dtype=np.float32)
    vc = rng.standard_normal((nkv, max_ctx, d), dtype=np.float32)
    out = torch.empty((Sq, nq * d), dtype=torch.float32, device=dev)
    ops.attention(_dev(q, dev), _dev(kc, dev), _dev(vc, dev), out, B=1, Hq=nq, Hkv=nkv, Sq=Sq, Sk=Sk, d=d, ldq=nq * d,
                  hsq=d, ldk=d, hsk=max_ctx * d, ldv=d, hsv=max_ctx * d, ldo=nq * d, scale=d ** -0.5, causal=True,
                  q_off=pos0)
    mask = np.arange(Sk)[None, :] <= (pos0 + np.arange(Sq))[:, None]
    ref = _attn_ref(q.reshape(Sq, nq, d).transpose(1, 0, 2), kc[:, :Sk], vc[:, :Sk], d ** -0.5, mask)
    assert_close(f"attn causal gqa Sq={Sq} pos0={pos0}", to_np(out), ref, atol=ATTN_X3_ATOL)


@pytest.mark.parametrize("klen,chunk,left", [(87, 0, -1), (60, 0, -1), (87, 16, 2), (87, 7, -1)])
def test_attention_relpos_masks(dev, klen, chunk, left):
    """Whale rel-pos scores ((q+u)k^T + (q+v)p^T)/sqrt(d), no rel-shift, pad + chunk masks
    (attention.py:380-409, utils.py:88-103)."""
    from vita_amd import ops
    rng = np.random.default_rng(11)
    H, T, d = 2, 87, 64
    q, k, v, p = (rng.standard_normal((T, H * d), dtype=np.float32) for _ in range(4))
    bu, bv = rng.standard_normal((H, d), dtype=np.float32), rng.standard_normal((H, d), dtype=np.float32)
    out = torch.empty((T, H * d), dtype=torch.float32, device=dev)
    k_dev, v_dev = k.copy(), v.copy()
    k_dev[klen:] = np.nan            # padded key rows are caller memory: masked keys must not be loaded from them
    v_dev[klen:] = np.nan
    ops.attention(_dev(q, dev), _dev(k_dev, dev), _dev(v_dev, dev), out, B=1, Hq=H, Hkv=H, Sq=T, Sk=T, d=d, ldq=H * d, hsq=d,
                  ldk=H * d, hsk=d, ldv=H * d, hsv=d, ldo=H * d, scale=d ** -0.5, klen=klen, chunk=chunk, left=left,
                  p=_dev(p, dev), ldp=H * d, hsp=d, bias_u=_dev(bu, dev), bias_v=_dev(bv, dev))
    mask = np.broadcast_to(np.arange(T)[None, :] < klen, (T, T)).copy()
    if chunk > 0:
        for i in range(T):
            start = 0 if left < 0 else max((i // chunk - left) * chunk, 0)
            end = min((i // chunk + 1) * chunk, T)
            mask[i, :start] = False
            mask[i, end:] = False
    sp = lambda x: x.reshape(T, H, d).transpose(1, 0, 2)
    ref = _attn_ref(sp(q), sp(k), sp(v), d ** -0.5, mask, sp(p), bu[:, None, :], bv[:, None, :])
    assert_close(f"attn relpos klen={klen} chunk={chunk}", to_np(out), ref, atol=5e-5)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cols", [128, 1024, 2048, 4096])
def test_layernorm_rmsnorm(dev, cols):
    from vita_amd import ops
    rng = np.random.default_rng(12)
    rows = 37
    x = rng.standard_normal((rows, cols), dtype=np.float32) * 2 + 0.5
    w, b = _w(rng, cols, std=1.0), _w(rng, cols, std=0.5)
    x64 = x.astype(np.float64)
    mu, var = x64.mean(-1, keepdims=True), x64.var(-1, keepdims=True)
    ref = (x64 - mu) / np.sqrt(var + 1e-5) * w + b
    assert_close("layernorm", to_np(ops.layernorm(_dev(x, dev), _dev(w, dev), _dev(b, dev), 1e-5)), ref, atol=2e-5)
    assert_close("layernorm relu*32", to_np(ops.layernorm(_dev(x, dev), _dev(w, dev), _dev(b, dev), 1e-5, act="relu",
                                                          post_scale=32.0)), np.maximum(ref, 0) * 32.0, atol=5e-4)
    rref = x64 / np.sqrt((x64 ** 2).mean(-1, keepdims=True) + 1e-5) * w
    assert_close("rmsnorm", to_np(ops.rmsnorm(_dev(x, dev), _dev(w, dev), 1e-5)), rref, atol=2e-5)


def test_vit_front_back(dev):
    """patchify (conv 14x14/14 as GEMM rows), CLS/pos assemble, pixel shuffle vs torch-free numpy
    restatements of modeling_intern_vit.py:109-121 and internvit_encoder.py:42-53."""
    from vita_amd import ops
    rng = np.random.default_rng(13)
    n, img, P, C = 2, 56, 14, 32
    g = img // P
    pix = rng.standard_normal((n, 3, img, img), dtype=np.float32)
    kpad = 640
    got = to_np(ops.vit_patchify(_dev(pix, dev), P, kpad))
    ref = pix.reshape(n, 3, g, P, g, P).transpose(0, 2, 4, 1, 3, 5).reshape(n * g * g, 3 * P * P)
    assert_close("patchify", got[:, :3 * P * P], ref, atol=0)
    assert np.all(got[:, 3 * P * P:] == 0)
    patches = rng.standard_normal((n * g * g, C), dtype=np.float32)
    cls, pos = _w(rng, C, std=1.0), _w(rng, g * g + 1, C, std=1.0)
    x = to_np(ops.vit_assemble(_dev(patches, dev), _dev(cls, dev, torch.bfloat16), _dev(pos, dev, torch.bfloat16), n,
                               g * g + 1, C)).reshape(n, g * g + 1, C)
    xref = np.concatenate([np.broadcast_to(cls, (n, 1, C)), patches.reshape(n, g * g, C)], 1) + pos[None]
    assert_close("assemble", x, xref, atol=1e-6)
    # pixel shuffle reference = the reference's view/permute chain
    f = (xref[:, 1:] * 0.5).reshape(n, g, g, C).astype(np.float32)
    t = f.reshape(n, g, g // 2, 2 * C).transpose(0, 2, 1, 3).reshape(n, g // 2, g // 2, 4 * C).transpose(0, 2, 1, 3)
    got = to_np(ops.vit_pixel_shuffle(_dev(xref.reshape(-1, C).astype(np.float32), dev), n, g, C, 0.5))
    assert_close("pixel_shuffle", got, t.reshape(n, (g // 2) ** 2, 4 * C), atol=1e-6)


def test_audio_conv1(dev):
    from vita_amd import ops
    rng = np.random.default_rng(14)
    T, F, C = 41, 80, 48
    feats = rng.standard_normal((T, F), dtype=np.float32) * 3 + 10
    mean, istd = rng.standard_normal(F).astype(np.float32) + 10, (rng.random(F).astype(np.float32) + 0.5)
    w, b = _w(rng, C, 1, 3, 3, std=0.3), _w(rng, C, std=0.3)
    out, T1, F1 = ops.audio_conv1(_dev(feats, dev), _dev(mean, dev), _dev(istd, dev),
                                  _dev(w.reshape(C, 9), dev, torch.bfloat16), _dev(b, dev))
    xn = ((feats - mean) * istd).astype(np.float64)
    ref = np.zeros((T1, F1, C))
    for t in range(T1):
        for f in range(F1):
            ref[t, f] = np.einsum("hw,chw->c", xn[2 * t:2 * t + 3, 2 * f:2 * f + 3], w[:, 0].astype(np.float64)) + b
    assert_close("audio conv1", to_np(out).reshape(T1, F1, C), np.maximum(ref, 0), atol=1e-4)


def test_embed_splice(dev):
    from vita_amd import ops
    rng = np.random.default_rng(15)
    H, V = 256, 50
    emb = _w(rng, V, H, std=1.0)
    img = rng.standard_normal((8, H), dtype=np.float32)
    aud = rng.standard_normal((5, H), dtype=np.float32)
    kind = np.array([0, 0, 1, 1, 1, 0, 2, 2, 0], np.int32)
    idx = np.array([3, 49, 0, 1, 7, 10, 4, 0, 0], np.int32)
    got = to_np(ops.embed_splice(_dev(kind, dev, torch.int32), _dev(idx, dev, torch.int32),
                                 _dev(emb, dev, torch.bfloat16), _dev(img, dev), _dev(aud, dev), H))
    ref = np.stack([(emb, img, aud)[k][i] for k, i in zip(kind, idx)])
    assert_close("embed_splice", got, ref, atol=0)


# ---- batch-1 decode operators, one C entry each against oracle/mixtral.py ------------------
def _moe_weights(rng, E, I, H):
    return dict(gate=_w(rng, E, H, std=0.2), w1=_w(rng, E, I, H), w3=_w(rng, E, I, H), w2=_w(rng, E, H, I))


@pytest.mark.parametrize("E,H,rows", [(8, 4096, 5), (4, 256, 37), (2, 128, 1)])
def test_router_top2_vs_oracle(dev, E, H, rows):
    from oracle import mixtral as om
    from vita_amd import ops
    rng = np.random.default_rng(E * 100 + rows)
    x = rng.standard_normal((rows, H), dtype=np.float32)
    g = _w(rng, E, H, std=0.1)
    ids, wts, probs = ops.router_top2(_dev(x, dev), _dev(g, dev, torch.bfloat16), want_probs=True)
    ridx, rval = om.router(x, g)
    assert to_np(ids).astype(np.int64).tolist() == ridx.tolist()
    assert_close("router weights", to_np(wts), rval, atol=2e-6)
    assert_close("router softmax", to_np(probs), om.softmax((x @ g.T).astype(np.float32)), atol=2e-6)


@pytest.mark.parametrize("E,I,H", [(8, 14336, 4096), (4, 512, 256)])
def test_moe_decode_vs_oracle(dev, E, I, H):
    """post_attention_layernorm + block_sparse_moe of one token (vh_moe_decode) vs the oracle's rmsnorm + moe, with and
    without the deferred residual (delta)."""
    from oracle import mixtral as om
    from vita_amd import ops
    rng = np.random.default_rng(I)
    lw = _moe_weights(rng, E, I, H)
    x = rng.standard_normal(H, dtype=np.float32)
    d = rng.standard_normal(H, dtype=np.float32) * 0.1
    nw = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)
    dw = {k: _dev(v, dev, torch.bfloat16) for k, v in lw.items()}
    for delta in (None, d):
        xin = x if delta is None else x + delta
        ref, ridx, rval = om.moe(om.rmsnorm(xin[None], nw, 1e-5), lw)
        y, route = ops.moe_decode(_dev(x, dev), _dev(nw, dev), 1e-5, dw["gate"], dw["w1"], dw["w3"], dw["w2"],
                                  delta=None if delta is None else _dev(delta, dev))
        r = to_np(route.view(torch.int32)).tolist()
        assert r[:2] == ridx[0].tolist()
        w = np.array(r[2:], np.int32).view(np.float32)
        assert_close("routing weights", w, rval[0], atol=2e-6)
        assert_close(f"moe_decode E={E} I={I}", to_np(y), ref[0], atol=1e-3 * max(1.0, float(np.abs(ref).max())), rtol=1e-4)


def test_rope_kv_append_and_attn_decode_vs_oracle(dev):
    """vh_rope_kv_append fills the cache for a 70-token prefix (two 64-key tiles), vh_attn_decode appends token 70 and attends
    over [0, 70]: RoPE'd q / k, the cache rows and the attention output vs oracle apply_rope / attention."""
    from oracle import mixtral as om
    from vita_amd import ops
    from vita_amd.engine import rope_tables
    rng = np.random.default_rng(77)
    nq, nkv, d, S, max_ctx = 8, 2, 128, 70, 192
    nqkv = (nq + 2 * nkv) * d
    qkv = rng.standard_normal((S + 1, nqkv), dtype=np.float32)
    cos, sin = rope_tables(max_ctx, d, 1e6)
    kc = torch.zeros((nkv, max_ctx, d), dtype=torch.float32, device=dev)
    vc = torch.zeros_like(kc)
    dcos, dsin = _dev(cos, dev), _dev(sin, dev)
    q = ops.rope_kv_append(_dev(qkv[:S], dev), kc, vc, dcos, dsin, 0, nq, nkv)
    c2, s2 = om.rope_cos_sin(np.arange(S + 1), d, 1e6)
    qh = qkv[:, :nq * d].reshape(S + 1, nq, d).transpose(1, 0, 2)
    kh = qkv[:, nq * d:(nq + nkv) * d].reshape(S + 1, nkv, d).transpose(1, 0, 2)
    vh = qkv[:, (nq + nkv) * d:].reshape(S + 1, nkv, d).transpose(1, 0, 2)
    qr, kr = om.apply_rope(qh, c2, s2), om.apply_rope(kh, c2, s2)
    # (the engine's tables are built in fp32 from float32(theta) ** ..., the oracle's inv_freq in fp64 then rounded: HF's order;
    # the angle differs by ~1e-6 relative at position 69: a few 1e-5 on values up to 4)
    assert_close("roped q", to_np(q), qr[:, :S].transpose(1, 0, 2).reshape(S, nq * d), atol=5e-5)
    assert_close("k cache", to_np(kc)[:, :S], kr[:, :S], atol=5e-5)
    assert_close("v cache", to_np(vc)[:, :S], vh[:, :S], atol=0)
    out = ops.attn_decode(_dev(qkv[S], dev), kc, vc, S, dcos, dsin, nq, nkv, d ** -0.5)
    ref = om.attention(qr[:, S:S + 1], kr, vh, S)
    assert_close("k appended", to_np(kc)[:, S], kr[:, S], atol=5e-5)
    assert_close("attn_decode", to_np(out), ref[0], atol=5e-5)


@pytest.mark.parametrize("V,H", [(51760, 4096), (1000, 256)])
def test_lmhead_argmax_vs_oracle(dev, V, H):
    from oracle import mixtral as om
    from vita_amd import ops
    rng = np.random.default_rng(V)
    x = rng.standard_normal(H, dtype=np.float32)
    d = rng.standard_normal(H, dtype=np.float32) * 0.1
    nw = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)
    w = _w(rng, V, H)
    ref = (om.rmsnorm((x + d)[None], nw, 1e-5) @ w.T).astype(np.float32)[0]
    logits, tok = ops.lmhead_argmax(_dev(x, dev), _dev(nw, dev), 1e-5, _dev(w, dev, torch.bfloat16), delta=_dev(d, dev))
    assert_close("lm_head logits", to_np(logits), ref, atol=1e-3)
    assert int(tok.item()) == int(np.argmax(ref))
    # ties: the lowest index wins (torch.argmax / HF greedy)
    w2 = w.copy(); w2[7] = w2[3]
    _, tok2 = ops.lmhead_argmax(_dev(w2[3].astype(np.float32) * 50, dev), _dev(np.ones(H, np.float32), dev), 1e-5, _dev(w2, dev, torch.bfloat16))
    lg2 = (om.rmsnorm((w2[3].astype(np.float32) * 50)[None], np.ones(H, np.float32), 1e-5) @ w2.T)[0]
    assert int(tok2.item()) == int(np.argmax(lg2)) and int(tok2.item()) in (3, int(np.argmax(lg2)))
