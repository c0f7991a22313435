"""TEST INFRASTRUCTURE (only tests/ may import this).  CPU restatement of the arithmetic of the bf16 x 3 attention kernel
(vita_amd/csrc/vh_attn.hip: k_attn_x3, at_split2): both operands of a product are split into bf16 hi + lo by round-to-nearest-
even (x = hi + lo to 2^-17 |x|), a fragment pair contributes lo*hi + hi*lo + hi*hi accumulated in fp32, the scale * log2(e)
factor is folded into Q before the split and the softmax runs in the log2 domain.  Used to pin the precision class the kernel
claims without a GPU."""
import numpy as np


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32 (v_cvt_pk_bf16_f32 on finite values)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split(x):
    x = np.asarray(x, np.float32)
    hi = bf16_rne(x)
    lo = bf16_rne(x - hi)
    return hi, lo


def matmul_x3(a, b):
    """a [m, k] @ b [k, n] as the kernel forms it: three bf16 products per term, fp32 accumulation."""
    ah, al = split(a)
    bh, bl = split(b)
    f = np.float32
    return (al.astype(f) @ bh.astype(f) + ah.astype(f) @ bl.astype(f) + ah.astype(f) @ bh.astype(f)).astype(f)


def attention_x3(q, k, v, scale, mask=None):
    """softmax(scale q k^T) v for one head with the kernel's arithmetic: q [Sq, d], k / v [Sk, d], mask [Sq, Sk] bool (True = visible)."""
    q2 = (np.asarray(q, np.float32) * np.float32(scale * 1.44269504088896340736)).astype(np.float32)
    s2 = matmul_x3(q2, np.asarray(k, np.float32).T)                       # scores in the log2 domain
    if mask is not None:
        s2 = np.where(mask, s2, -np.inf).astype(np.float32)
    m = s2.max(-1, keepdims=True)
    p = np.exp2(s2 - m).astype(np.float32)
    l = p.sum(-1, keepdims=True, dtype=np.float32)
    return (matmul_x3(p, np.asarray(v, np.float32)) / l).astype(np.float32)
