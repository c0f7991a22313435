// vh_decode.hip — batch-1 decode kernels for the Mixtral backbone.
//
// One generated token touches every active weight exactly once, so decode is a pure
// HBM stream (25.66 GB/token at TP=1).  Five kernels per layer, each a row-parallel
// GEMV whose prologue recomputes the (tiny, L2-resident) activation-side work instead
// of paying a kernel boundary for it:
//
//   k_dec_qkv     x = x_in + delta ; RMSNorm ; fused Q|K|V GEMV           (K20,K21)
//   k_dec_attn    RoPE(q,k_new) ; KV-cache append ; split-KV GQA attention (K22,K23)
//   k_dec_oproj   split-KV combine ; O-projection GEMV -> delta_attn       (K24)
//   k_dec_gateup  x = x_in + delta ; RMSNorm ; router softmax/top-2 (on device, no
//                 host sync) ; gate|up GEMV of the two chosen experts ; SiLU*up  (K25,K26)
//   k_dec_down    down GEMV of the two experts, routing-weighted sum -> delta_moe (K26)
//   k_dec_lmhead  final RMSNorm ; LM-head GEMV ; per-block argmax          (K27)
//   k_dec_select  global argmax ; append token ; pos++ ; next embedding    (K28,K18)
//
// GEMV shape: a 256-thread block owns R output rows; thread t owns 16-byte chunks
// c = t + 256*j of every row (a wave reads 1 KiB contiguous per load instruction),
// keeps its slice of the fp32 activation vector in registers and issues all R*NJ
// weight loads before the first FMA (deep vmcnt, no LDS round trip — guide §5 "GEMV /
// M<=16 decode weights").  Output rows are reduced wave->LDS->thread, no atomics, so
// results are deterministic.  Residual adds are deferred into the consumer's prologue
// ("delta" buffers) so that under tensor parallelism the same kernels run unchanged
// with an all-reduce on the delta buffer between them.
//
// Reference semantics restated: transformers/models/mixtral/modeling_mixtral.py
// (MixtralRMSNorm, MixtralAttention + apply_rotary_pos_emb, MixtralTopKRouter,
// MixtralExperts) as called from vita/model/language_model/vita_mixtral.py:158-173.
#include "vh_common.h"
#include "vh_kernels.h"

namespace {

// ---- activation slice handling ------------------------------------------------------
template <int NJ>
__device__ __forceinline__ void load_x(const float* __restrict__ x, int K, float (&xr)[NJ][8]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c * 8 < K) {
            const float4 a = reinterpret_cast<const float4*>(x)[c * 2];
            const float4 b = reinterpret_cast<const float4*>(x)[c * 2 + 1];
            xr[j][0] = a.x; xr[j][1] = a.y; xr[j][2] = a.z; xr[j][3] = a.w;
            xr[j][4] = b.x; xr[j][5] = b.y; xr[j][6] = b.z; xr[j][7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) xr[j][i] = 0.f;
        }
    }
}

template <int NJ>
__device__ __forceinline__ void add_x(const float* __restrict__ d, int K, float (&xr)[NJ][8]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c * 8 < K) {
            const float4 a = reinterpret_cast<const float4*>(d)[c * 2];
            const float4 b = reinterpret_cast<const float4*>(d)[c * 2 + 1];
            xr[j][0] += a.x; xr[j][1] += a.y; xr[j][2] += a.z; xr[j][3] += a.w;
            xr[j][4] += b.x; xr[j][5] += b.y; xr[j][6] += b.z; xr[j][7] += b.w;
        }
    }
}

template <int NJ>
__device__ __forceinline__ void store_x(float* __restrict__ x, int K, const float (&xr)[NJ][8]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c * 8 < K) {
            reinterpret_cast<float4*>(x)[c * 2] = make_float4(xr[j][0], xr[j][1], xr[j][2], xr[j][3]);
            reinterpret_cast<float4*>(x)[c * 2 + 1] = make_float4(xr[j][4], xr[j][5], xr[j][6], xr[j][7]);
        }
    }
}

// MixtralRMSNorm in fp32: (x * rsqrt(mean(x^2) + eps)) * w   (modeling_mixtral.py:143-148)
template <int NJ>
__device__ __forceinline__ void rmsnorm_x(const float* __restrict__ w, int K, float eps,
                                          float (&xr)[NJ][8], float* red) {
    float ss[1] = {0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) ss[0] = fmaf(xr[j][i], xr[j][i], ss[0]);
    block256_sum<1>(ss, red);
    const float inv = rsqrtf(ss[0] / (float)K + eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c * 8 < K) {
            const float4 a = reinterpret_cast<const float4*>(w)[c * 2];
            const float4 b = reinterpret_cast<const float4*>(w)[c * 2 + 1];
            xr[j][0] = xr[j][0] * inv * a.x; xr[j][1] = xr[j][1] * inv * a.y;
            xr[j][2] = xr[j][2] * inv * a.z; xr[j][3] = xr[j][3] * inv * a.w;
            xr[j][4] = xr[j][4] * inv * b.x; xr[j][5] = xr[j][5] * inv * b.y;
            xr[j][6] = xr[j][6] * inv * b.z; xr[j][7] = xr[j][7] * inv * b.w;
        }
    }
}

// R rows of W (bf16, row stride ldw elements) dotted with the register-resident x.
// rows[r] must be valid pointers (callers clamp out-of-range rows and drop the result).
template <int NJ, int R>
__device__ __forceinline__ void gemv_rows(const uint16_t* const (&rows)[R], int K,
                                          const float (&xr)[NJ][8], float (&acc)[R]) {
    uint4 w[R][NJ];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = threadIdx.x + j * 256;
            if (c * 8 < K) w[r][j] = ld_weight16(rows[r] + (size_t)c * 8);
            else w[r][j] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) a += dot8_bf16_f32(w[r][j], xr[j]);
        acc[r] = a;
    }
}

// ---- K_A: residual add + RMSNorm + fused QKV GEMV -----------------------------------
template <int NJ, int R>
__global__ __launch_bounds__(256) void k_dec_qkv(const float* __restrict__ x_in, const float* __restrict__ delta,
                                                 float* __restrict__ x_out, const float* __restrict__ norm_w,
                                                 float eps, const uint16_t* __restrict__ W, int N, int K,
                                                 float* __restrict__ out) {
    __shared__ float red[4 * R];
    float xr[NJ][8];
    load_x<NJ>(x_in, K, xr);
    if (delta) add_x<NJ>(delta, K, xr);
    if (blockIdx.x == 0 && x_out) store_x<NJ>(x_out, K, xr);
    rmsnorm_x<NJ>(norm_w, K, eps, xr, red);

    const int n0 = blockIdx.x * R;
    const uint16_t* rows[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rows[r] = W + (size_t)min(n0 + r, N - 1) * K;
    float acc[R];
    gemv_rows<NJ, R>(rows, K, xr, acc);
    block256_sum<R>(acc, red);
    if (threadIdx.x < R && n0 + threadIdx.x < N) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) if (threadIdx.x == r) v = acc[r];
        out[n0 + threadIdx.x] = v;
    }
}

// ---- K_B: RoPE + KV append + split-KV GQA attention ---------------------------------
// grid (nkv, nsplit); wave w < G serves query head h*G + w; lanes index keys for QK^T
// and head dims (2 per lane) for PV.  K/V tiles (64 keys) are staged through LDS once
// and shared by the G query heads of the KV head (K23: "4 q-heads share each LDS-staged
// KV tile").  head_dim is 128 (config.json:16-44).
#define DA_KT 64
#define DA_KSTR 132
__global__ __launch_bounds__(256) void k_dec_attn(const float* __restrict__ qkv, float* __restrict__ kcache,
                                                  float* __restrict__ vcache, const int* __restrict__ pos_ptr,
                                                  const float* __restrict__ rope_cos,
                                                  const float* __restrict__ rope_sin, float* __restrict__ part_o,
                                                  float* __restrict__ part_ml, int nq, int nkv, int max_ctx,
                                                  int nsplit, float scale) {
    __shared__ __attribute__((aligned(16))) float q_s[4][128];
    __shared__ __attribute__((aligned(16))) float kn_s[128];
    __shared__ __attribute__((aligned(16))) float vn_s[128];
    __shared__ __attribute__((aligned(16))) float Kt[DA_KT * DA_KSTR];
    __shared__ __attribute__((aligned(16))) float Vt[DA_KT * 128];

    const int h = blockIdx.x, sp = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int G = nq / nkv;
    const int pos = *pos_ptr;
    const int ctx = pos + 1;
    const int span = (((ctx + nsplit - 1) / nsplit) + DA_KT - 1) & ~(DA_KT - 1);
    const int k0 = sp * span;
    const int k1 = min(ctx, k0 + span);
    const int head = h * G + wid;

    if (k0 >= k1) {  // empty split (uniform per block)
        if (wid < G) {
            reinterpret_cast<float2*>(part_o + ((size_t)head * nsplit + sp) * 128)[lane] = make_float2(0.f, 0.f);
            if (lane == 0) {
                part_ml[((size_t)head * nsplit + sp) * 2] = -INFINITY;
                part_ml[((size_t)head * nsplit + sp) * 2 + 1] = 0.f;
            }
        }
        return;
    }

    // rotate-half RoPE (modeling_mixtral.py:203-241): out = x*cos + rotate_half(x)*sin
    const float c = rope_cos[(size_t)pos * 64 + lane], s = rope_sin[(size_t)pos * 64 + lane];
    if (wid < G) {
        const float a = qkv[head * 128 + lane], b = qkv[head * 128 + 64 + lane];
        q_s[wid][lane] = a * c - b * s;
        q_s[wid][lane + 64] = b * c + a * s;
    }
    const bool has_new = (pos >= k0) && (pos < k1);
    if (has_new && wid == 3) {  // wave 3 is idle for G<4 and cheap otherwise
        const float* kp = qkv + (size_t)nq * 128 + h * 128;
        const float* vp = qkv + (size_t)(nq + nkv) * 128 + h * 128;
        const float a = kp[lane], b = kp[lane + 64];
        const float ka = a * c - b * s, kb = b * c + a * s;
        const float va = vp[lane], vb = vp[lane + 64];
        kn_s[lane] = ka; kn_s[lane + 64] = kb;
        vn_s[lane] = va; vn_s[lane + 64] = vb;
        float* kc = kcache + ((size_t)h * max_ctx + pos) * 128;
        float* vc = vcache + ((size_t)h * max_ctx + pos) * 128;
        kc[lane] = ka; kc[lane + 64] = kb;
        vc[lane] = va; vc[lane + 64] = vb;
    }
    __syncthreads();

    float4 qreg[32];
    if (wid < G) {
#pragma unroll
        for (int i = 0; i < 32; ++i) qreg[i] = reinterpret_cast<const float4*>(q_s[wid])[i];
    }

    float m = -INFINITY, l = 0.f;
    float2 o = make_float2(0.f, 0.f);

    for (int t0 = k0; t0 < k1; t0 += DA_KT) {
        // stage K and V tiles: 64 rows x 32 float4 each
        for (int idx = tid; idx < DA_KT * 32; idx += 256) {
            const int row = idx >> 5, c4 = idx & 31;
            const int key = t0 + row;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (key < k1) {
                if (key == pos) {
                    kv = reinterpret_cast<const float4*>(kn_s)[c4];
                    vv = reinterpret_cast<const float4*>(vn_s)[c4];
                } else {
                    kv = reinterpret_cast<const float4*>(kcache + ((size_t)h * max_ctx + key) * 128)[c4];
                    vv = reinterpret_cast<const float4*>(vcache + ((size_t)h * max_ctx + key) * 128)[c4];
                }
            }
            *reinterpret_cast<float4*>(&Kt[row * DA_KSTR + c4 * 4]) = kv;
            *reinterpret_cast<float4*>(&Vt[row * 128 + c4 * 4]) = vv;
        }
        __syncthreads();
        if (wid < G) {
            const bool valid = (t0 + lane) < k1;
            float sc = 0.f;
            const float4* kr = reinterpret_cast<const float4*>(&Kt[lane * DA_KSTR]);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float4 kk = kr[i];
                sc = fmaf(qreg[i].x, kk.x, sc);
                sc = fmaf(qreg[i].y, kk.y, sc);
                sc = fmaf(qreg[i].z, kk.z, sc);
                sc = fmaf(qreg[i].w, kk.w, sc);
            }
            sc = valid ? sc * scale : -INFINITY;
            const float m_new = fmaxf(m, wave_max(sc));  // finite: the tile has >= 1 valid key
            const float alpha = __expf(m - m_new);       // m = -inf -> 0
            const float p = valid ? __expf(sc - m_new) : 0.f;
            l = l * alpha + wave_sum(p);
            o.x *= alpha; o.y *= alpha;
#pragma unroll
            for (int kk = 0; kk < DA_KT; ++kk) {
                const float pk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), kk));
                const float2 vv = reinterpret_cast<const float2*>(&Vt[kk * 128])[lane];
                o.x = fmaf(pk, vv.x, o.x);
                o.y = fmaf(pk, vv.y, o.y);
            }
            m = m_new;
        }
        __syncthreads();
    }
    if (wid < G) {
        reinterpret_cast<float2*>(part_o + ((size_t)head * nsplit + sp) * 128)[lane] = o;
        if (lane == 0) {
            part_ml[((size_t)head * nsplit + sp) * 2] = m;
            part_ml[((size_t)head * nsplit + sp) * 2 + 1] = l;
        }
    }
}

// ---- K_C: split-KV combine + O-projection GEMV --------------------------------------
// K = nq*128 (this rank's heads); chunk c covers dims [8c, 8c+8) of head c>>4.
template <int NJ, int R>
__global__ __launch_bounds__(256) void k_dec_oproj(const float* __restrict__ part_o,
                                                   const float* __restrict__ part_ml, int nsplit,
                                                   const uint16_t* __restrict__ W, int N, int K,
                                                   float* __restrict__ out) {
    __shared__ float red[4 * R];
    float xr[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = threadIdx.x + j * 256;
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[j][i] = 0.f;
        if (c * 8 < K) {
            const int head = c >> 4, d0 = (c & 15) * 8;
            float M = -INFINITY;
            for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_ml[((size_t)head * nsplit + s) * 2]);
            float den = 0.f;
            for (int s = 0; s < nsplit; ++s) {
                const float ms = part_ml[((size_t)head * nsplit + s) * 2];
                const float ls = part_ml[((size_t)head * nsplit + s) * 2 + 1];
                const float wgt = (ms == -INFINITY) ? 0.f : __expf(ms - M);
                den = fmaf(wgt, ls, den);
                const float4 a = reinterpret_cast<const float4*>(part_o + ((size_t)head * nsplit + s) * 128 + d0)[0];
                const float4 b = reinterpret_cast<const float4*>(part_o + ((size_t)head * nsplit + s) * 128 + d0)[1];
                xr[j][0] = fmaf(wgt, a.x, xr[j][0]); xr[j][1] = fmaf(wgt, a.y, xr[j][1]);
                xr[j][2] = fmaf(wgt, a.z, xr[j][2]); xr[j][3] = fmaf(wgt, a.w, xr[j][3]);
                xr[j][4] = fmaf(wgt, b.x, xr[j][4]); xr[j][5] = fmaf(wgt, b.y, xr[j][5]);
                xr[j][6] = fmaf(wgt, b.z, xr[j][6]); xr[j][7] = fmaf(wgt, b.w, xr[j][7]);
            }
            const float inv = 1.0f / den;
#pragma unroll
            for (int i = 0; i < 8; ++i) xr[j][i] *= inv;
        }
    }
    const int n0 = blockIdx.x * R;
    const uint16_t* rows[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rows[r] = W + (size_t)min(n0 + r, N - 1) * K;
    float acc[R];
    gemv_rows<NJ, R>(rows, K, xr, acc);
    block256_sum<R>(acc, red);
    if (threadIdx.x < R && n0 + threadIdx.x < N) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) if (threadIdx.x == r) v = acc[r];
        out[n0 + threadIdx.x] = v;
    }
}

// ---- K_D: residual add + RMSNorm + router + gate|up GEMV + SiLU*up ------------------
// Router (modeling_mixtral.py:96-111): logits = x_n @ Wg^T ; softmax fp32 ; top-2 ;
// renormalise.  Every block recomputes it (8 rows, L2-resident) so the expert ids never
// leave the device.  route_out = {e0, e1, bits(w0), bits(w1)}.
#define GU_RP 4  // (gate,up) row pairs per block iteration
template <int NJ>
__global__ __launch_bounds__(256) void k_dec_gateup(const float* __restrict__ x_in, const float* __restrict__ delta,
                                                    float* __restrict__ x_out, const float* __restrict__ norm_w,
                                                    float eps, const uint16_t* __restrict__ Wg, int E,
                                                    const uint16_t* __restrict__ W1, const uint16_t* __restrict__ W3,
                                                    int I, int K, int* __restrict__ route_out,
                                                    float* __restrict__ hbuf) {
    __shared__ float red[4 * 8];
    float xr[NJ][8];
    load_x<NJ>(x_in, K, xr);
    if (delta) add_x<NJ>(delta, K, xr);
    if (blockIdx.x == 0 && x_out) store_x<NJ>(x_out, K, xr);
    rmsnorm_x<NJ>(norm_w, K, eps, xr, red);

    // router
    float lg[8];
    {
        const uint16_t* rows[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) rows[e] = Wg + (size_t)min(e, E - 1) * K;
        gemv_rows<NJ, 8>(rows, K, xr, lg);
        block256_sum<8>(lg, red);
    }
    int e0 = 0, e1 = 0;
    float w0, w1;
    {
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < E) mx = fmaxf(mx, lg[e]);
        float pr[8], sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { pr[e] = (e < E) ? expf(lg[e] - mx) : 0.f; sum += pr[e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) pr[e] = pr[e] / sum;
        float b0 = -1.f, b1 = -1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < E && pr[e] > b0) { b0 = pr[e]; e0 = e; }
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < E && e != e0 && pr[e] > b1) { b1 = pr[e]; e1 = e; }
        const float t = b0 + b1;
        w0 = b0 / t; w1 = b1 / t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        route_out[0] = e0; route_out[1] = e1;
        route_out[2] = __float_as_int(w0); route_out[3] = __float_as_int(w1);
    }

    const int per_slot = I / GU_RP;
    const int n_iter = 2 * per_slot;
    for (int it = blockIdx.x; it < n_iter; it += gridDim.x) {
        const int slot = it / per_slot;
        const int i0 = (it - slot * per_slot) * GU_RP;
        const int e = slot ? e1 : e0;
        const uint16_t* rows[2 * GU_RP];
#pragma unroll
        for (int r = 0; r < GU_RP; ++r) {
            rows[r] = W1 + ((size_t)e * I + i0 + r) * K;
            rows[GU_RP + r] = W3 + ((size_t)e * I + i0 + r) * K;
        }
        float acc[2 * GU_RP];
        gemv_rows<NJ, 2 * GU_RP>(rows, K, xr, acc);
        block256_sum<2 * GU_RP>(acc, red);
        if (threadIdx.x < GU_RP) {
            float g = 0.f, u = 0.f;
#pragma unroll
            for (int r = 0; r < GU_RP; ++r) if (threadIdx.x == r) { g = acc[r]; u = acc[GU_RP + r]; }
            hbuf[(size_t)slot * I + i0 + threadIdx.x] = silu_f(g) * u;
        }
    }
}

// ---- K_E: down GEMV of both experts, routing-weighted sum ---------------------------
template <int NJ, int R>
__global__ __launch_bounds__(256) void k_dec_down(const float* __restrict__ hbuf, const int* __restrict__ route,
                                                  const uint16_t* __restrict__ W2, int N, int I,
                                                  float* __restrict__ out) {
    __shared__ float red[4 * R];
    const int e0 = route[0], e1 = route[1];
    const float w0 = __int_as_float(route[2]), w1 = __int_as_float(route[3]);
    const int n0 = blockIdx.x * R;
    float tot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) tot[r] = 0.f;
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        const int e = slot ? e1 : e0;
        const float wt = slot ? w1 : w0;
        float xr[NJ][8];
        load_x<NJ>(hbuf + (size_t)slot * I, I, xr);
        const uint16_t* rows[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = W2 + ((size_t)e * N + min(n0 + r, N - 1)) * I;
        float acc[R];
        gemv_rows<NJ, R>(rows, I, xr, acc);
#pragma unroll
        for (int r = 0; r < R; ++r) tot[r] = fmaf(wt, acc[r], tot[r]);
    }
    // sum over threads is linear, so the routing weights were applied per thread above
    block256_sum<R>(tot, red);
    if (threadIdx.x < R && n0 + threadIdx.x < N) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) if (threadIdx.x == r) v = tot[r];
        out[n0 + threadIdx.x] = v;
    }
}

// ---- K_F: final RMSNorm + LM head GEMV + per-block argmax ---------------------------
#define LM_R 8
template <int NJ>
__global__ __launch_bounds__(256) void k_dec_lmhead(const float* __restrict__ x_in, const float* __restrict__ delta,
                                                    const float* __restrict__ norm_w, float eps,
                                                    const uint16_t* __restrict__ W, int V, int K,
                                                    float* __restrict__ logits, float* __restrict__ blk_val,
                                                    int* __restrict__ blk_idx, const int* __restrict__ ngen_ptr,
                                                    int hist_rows) {
    __shared__ float red[4 * LM_R];
    // logits history: row = index of the token this step produces (clamped), so the host can
    // read every step's scores after a multi-step launch (HF generate(output_scores=True)).
    if (hist_rows > 1) logits += (size_t)min(*ngen_ptr, hist_rows - 1) * V;
    __shared__ float bv_s[LM_R];
    __shared__ int bi_s[LM_R];
    float xr[NJ][8];
    load_x<NJ>(x_in, K, xr);
    if (delta) add_x<NJ>(delta, K, xr);
    rmsnorm_x<NJ>(norm_w, K, eps, xr, red);
    float best = -INFINITY;
    int besti = 0x7fffffff;
    const int n_iter = (V + LM_R - 1) / LM_R;
    for (int it = blockIdx.x; it < n_iter; it += gridDim.x) {
        const int n0 = it * LM_R;
        const uint16_t* rows[LM_R];
#pragma unroll
        for (int r = 0; r < LM_R; ++r) rows[r] = W + (size_t)min(n0 + r, V - 1) * K;
        float acc[LM_R];
        gemv_rows<NJ, LM_R>(rows, K, xr, acc);
        block256_sum<LM_R>(acc, red);
        if (threadIdx.x < LM_R && n0 + threadIdx.x < V) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < LM_R; ++r) if (threadIdx.x == r) v = acc[r];
            logits[n0 + threadIdx.x] = v;
            if (v > best) { best = v; besti = n0 + threadIdx.x; }  // ascending n: first max wins
        }
    }
    if (threadIdx.x < LM_R) { bv_s[threadIdx.x] = best; bi_s[threadIdx.x] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = bv_s[0]; int bi = bi_s[0];
        for (int r = 1; r < LM_R; ++r)
            if (bv_s[r] > b || (bv_s[r] == b && bi_s[r] < bi)) { b = bv_s[r]; bi = bi_s[r]; }
        blk_val[blockIdx.x] = b; blk_idx[blockIdx.x] = bi;
    }
}

// ---- K_G: global argmax (lowest index on ties, as torch.argmax), bookkeeping, and the
// next step's input embedding (vita_arch.py:155-175 decode early-exit + embed_tokens).
__global__ __launch_bounds__(256) void k_dec_select(const float* __restrict__ blk_val, const int* __restrict__ blk_idx,
                                                    int nblk, const uint16_t* __restrict__ embed, int H,
                                                    float* __restrict__ x_next, int* __restrict__ pos_ptr,
                                                    int* __restrict__ ngen_ptr, int* __restrict__ out_tokens,
                                                    int max_out, int mode, int set_pos) {
    __shared__ float v_s[256];
    __shared__ int i_s[256];
    float b = -INFINITY; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        const float v = blk_val[i]; const int ix = blk_idx[i];
        if (v > b || (v == b && ix < bi)) { b = v; bi = ix; }
    }
    v_s[threadIdx.x] = b; i_s[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float v = v_s[threadIdx.x + s]; const int ix = i_s[threadIdx.x + s];
            if (v > v_s[threadIdx.x] || (v == v_s[threadIdx.x] && ix < i_s[threadIdx.x])) {
                v_s[threadIdx.x] = v; i_s[threadIdx.x] = ix;
            }
        }
        __syncthreads();
    }
    const int tok = i_s[0];
    if (threadIdx.x == 0) {
        // mode 1: decode step (append, pos++); mode 0: end of prefill (first token, pos = set_pos)
        const int n = mode ? *ngen_ptr : 0;
        if (n < max_out) out_tokens[n] = tok;
        *ngen_ptr = n + 1;
        *pos_ptr = mode ? (*pos_ptr + 1) : set_pos;
    }
    for (int c = threadIdx.x; c * 8 < H; c += 256) {
        const uint4 w = reinterpret_cast<const uint4*>(embed + (size_t)tok * H)[c];
        reinterpret_cast<float4*>(x_next)[c * 2] =
            make_float4(bf16_lo_to_f32(w.x), bf16_hi_to_f32(w.x), bf16_lo_to_f32(w.y), bf16_hi_to_f32(w.y));
        reinterpret_cast<float4*>(x_next)[c * 2 + 1] =
            make_float4(bf16_lo_to_f32(w.z), bf16_hi_to_f32(w.z), bf16_lo_to_f32(w.w), bf16_hi_to_f32(w.w));
    }
}

template <typename F>
inline int pick_nj(int K, F&& f) {
    // chunk slots per thread: K <= NJ * 2048
    if (K <= 2048) return f(std::integral_constant<int, 1>{});
    if (K <= 4096) return f(std::integral_constant<int, 2>{});
    if (K <= 8192) return f(std::integral_constant<int, 4>{});
    if (K <= 14336) return f(std::integral_constant<int, 7>{});
    return -1;
}

}  // namespace

// ---- launchers (declared in vh_kernels.h) -------------------------------------------
int vhk_dec_qkv(hipStream_t st, const float* x_in, const float* delta, float* x_out, const float* norm_w, float eps,
                const uint16_t* W, int N, int K, float* out) {
    constexpr int R = 4;
    return pick_nj(K, [&](auto nj) {
        hipLaunchKernelGGL((k_dec_qkv<decltype(nj)::value, R>), dim3((N + R - 1) / R), dim3(256), 0, st, x_in, delta,
                           x_out, norm_w, eps, W, N, K, out);
        return 0;
    });
}

int vhk_dec_attn(hipStream_t st, const float* qkv, float* kcache, float* vcache, const int* pos_ptr,
                 const float* rope_cos, const float* rope_sin, float* part_o, float* part_ml, int nq, int nkv,
                 int max_ctx, int nsplit, float scale) {
    if (nq % nkv != 0 || nq / nkv > 4) return -1;
    hipLaunchKernelGGL(k_dec_attn, dim3(nkv, nsplit), dim3(256), 0, st, qkv, kcache, vcache, pos_ptr, rope_cos,
                       rope_sin, part_o, part_ml, nq, nkv, max_ctx, nsplit, scale);
    return 0;
}

int vhk_dec_oproj(hipStream_t st, const float* part_o, const float* part_ml, int nsplit, const uint16_t* W, int N,
                  int K, float* out) {
    constexpr int R = 8;
    return pick_nj(K, [&](auto nj) {
        hipLaunchKernelGGL((k_dec_oproj<decltype(nj)::value, R>), dim3((N + R - 1) / R), dim3(256), 0, st, part_o,
                           part_ml, nsplit, W, N, K, out);
        return 0;
    });
}

int vhk_dec_gateup(hipStream_t st, const float* x_in, const float* delta, float* x_out, const float* norm_w, float eps,
                   const uint16_t* Wg, int E, const uint16_t* W1, const uint16_t* W3, int I, int K, int* route_out,
                   float* hbuf, int grid) {
    if (E > 8 || E < 2 || I % GU_RP != 0) return -1;
    const int n_iter = 2 * (I / GU_RP);
    if (grid <= 0 || grid > n_iter) grid = n_iter < 1024 ? n_iter : 1024;
    return pick_nj(K, [&](auto nj) {
        hipLaunchKernelGGL((k_dec_gateup<decltype(nj)::value>), dim3(grid), dim3(256), 0, st, x_in, delta, x_out,
                           norm_w, eps, Wg, E, W1, W3, I, K, route_out, hbuf);
        return 0;
    });
}

int vhk_dec_down(hipStream_t st, const float* hbuf, const int* route, const uint16_t* W2, int N, int I, float* out) {
    constexpr int R = 2;
    return pick_nj(I, [&](auto nj) {
        hipLaunchKernelGGL((k_dec_down<decltype(nj)::value, R>), dim3((N + R - 1) / R), dim3(256), 0, st, hbuf, route,
                           W2, N, I, out);
        return 0;
    });
}

int vhk_dec_lmhead(hipStream_t st, const float* x_in, const float* delta, const float* norm_w, float eps,
                   const uint16_t* W, int V, int K, float* logits, float* blk_val, int* blk_idx, int grid,
                   const int* ngen_ptr, int hist_rows) {
    return pick_nj(K, [&](auto nj) {
        hipLaunchKernelGGL((k_dec_lmhead<decltype(nj)::value>), dim3(grid), dim3(256), 0, st, x_in, delta, norm_w, eps,
                           W, V, K, logits, blk_val, blk_idx, ngen_ptr, hist_rows);
        return 0;
    });
}

int vhk_dec_select(hipStream_t st, const float* blk_val, const int* blk_idx, int nblk, const uint16_t* embed, int H,
                   float* x_next, int* pos_ptr, int* ngen_ptr, int* out_tokens, int max_out, int mode, int set_pos) {
    hipLaunchKernelGGL(k_dec_select, dim3(1), dim3(256), 0, st, blk_val, blk_idx, nblk, embed, H, x_next, pos_ptr,
                       ngen_ptr, out_tokens, max_out, mode, set_pos);
    return 0;
}
