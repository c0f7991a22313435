"""CPU pin of the numeric contract of the bf16 x 3 attention (vh_attn.hip k_attn_x3): the restated arithmetic
(oracle/bf16x3.py) against fp64 — split exactness, per-product error class, end-to-end attention error at the two head sizes,
and that dropping the third product (a plain bf16 x 2 scheme) would NOT meet the bar the GPU tests use."""
import numpy as np
import pytest

from oracle import bf16x3 as x3


def test_split_is_exact_to_2_pow_minus_17():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-8, 8, 200000))).astype(np.float32)
    hi, lo = x3.split(x)
    assert np.all(np.abs(x.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64)) <= 2.0 ** -17 * np.abs(x) * 1.0001)
    # hi and lo are bf16 values: the low 16 bits of their fp32 patterns are zero
    assert not np.any(hi.view(np.uint32) & 0xFFFF) and not np.any(lo.view(np.uint32) & 0xFFFF)


@pytest.mark.parametrize("d", [64, 128])
def test_attention_error_class(d):
    rng = np.random.default_rng(d)
    Sq, Sk = 48, 333
    q, k, v = (rng.standard_normal((n, d)).astype(np.float32) for n in (Sq, Sk, Sk))
    mask = np.arange(Sk)[None, :] <= (Sk - Sq + np.arange(Sq))[:, None]                  # causal with an offset, as a chunked prefill
    s = (q.astype(np.float64) @ k.astype(np.float64).T) * d ** -0.5
    s = np.where(mask, s, -np.inf)
    p = np.exp(s - s.max(-1, keepdims=True))
    ref = (p / p.sum(-1, keepdims=True)) @ v.astype(np.float64)
    got = x3.attention_x3(q, k, v, d ** -0.5, mask)
    err = np.abs(got - ref).max()
    print(f"d = {d}: bf16 x 3 attention vs fp64: max error {err:.2e}")
    assert err < 1e-4                                                                      # = ATTN_X3_ATOL of tests/test_ops_gpu.py
    # without the cross terms (hi * hi only) the error is two orders of magnitude larger: the third product is what buys the class
    qh, kh, vh = x3.bf16_rne(q), x3.bf16_rne(k), x3.bf16_rne(v)
    s1 = np.where(mask, (qh.astype(np.float64) @ kh.astype(np.float64).T) * d ** -0.5, -np.inf)
    p1 = np.exp(s1 - s1.max(-1, keepdims=True))
    plain = (x3.bf16_rne((p1 / p1.sum(-1, keepdims=True)).astype(np.float32)).astype(np.float64)) @ vh.astype(np.float64)
    assert np.abs(plain - ref).max() > 30 * err


def test_products_match_the_exact_mode_gemm_class():
    """one operand exact in bf16 (a weight), the other split: the GEMMs' two-product scheme; both split: three products —
    the same 2^-17-per-term class."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal((64, 4096)).astype(np.float32)
    b = rng.standard_normal((4096, 32)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    e3 = np.abs(x3.matmul_x3(a, b) - ref).max()
    ah, al = x3.split(a)
    bw = x3.bf16_rne(b)                                                                    # a bf16 weight matrix
    e2 = np.abs((al @ bw + ah @ bw).astype(np.float64) - a.astype(np.float64) @ bw.astype(np.float64)).max()
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    print(f"three-product error {e3:.2e}, two-product (bf16 weights) {e2:.2e}, sum|a||b| up to {scale.max():.0f}")
    assert e3 < 3 * 2.0 ** -17 * scale.max() and e2 < 3 * 2.0 ** -17 * scale.max()
