// vh_attn.hip — multi-query-row attention in exact fp32 on the matrix cores
// (v_mfma_f32_16x16x4_f32, gfx950 keeps an f32-in/f32-acc MFMA at the vector rate).
// Serves three call sites:
//   K5  InternViT global attention, 16 heads x 64, N=1025, no mask
//       (internvit/modeling_intern_vit.py:158-177)
//   K14 Whale rel-pos attention, 16 heads x 64: scores = ((q+u)k^T + (q+v)p^T)/sqrt(d),
//       NO rel-shift, pad/chunk mask -> excluded keys (whale/module/layer/attention.py:370-419,
//       whale/utils.py:88-146)
//   K23 Mixtral prefill: causal GQA, 32 q-heads / 8 kv-heads x 128, K/V read from the fp32
//       KV cache (HF eager_attention_forward, modeling_mixtral.py:256-279)
// The FLOP count of all three is small next to the GEMMs (53 GF for a 450-token prefill,
// 103 GF per ViT tile), so exact fp32 costs <1-2 ms and removes attention from the
// parity error budget.
//
// Structure: grid (ceil(Sq/64), Hq, B); 4 waves, each owning 16 query rows; K/V (and P)
// tiles of 32 keys staged through LDS and shared by the 4 waves; flash-style online
// softmax in the MFMA D layout (row = (lane>>4)*4 + r, col = lane&15) with 16-lane
// shuffles; probabilities go D-layout -> A-layout through a per-wave LDS patch.
// LDS row strides are chosen for the ds_read_b32 32-bank model: K/P stride D+2 (B-operand
// read of K^T: 16 rows x {k,k+1}), V stride D+16 (B-operand read: 2 rows x 16 cols).
#include "vh_common.h"
#include "vh_kernels.h"

namespace {

#define AT_KT 32
#define AT_PSTR 34

__device__ __forceinline__ bool key_visible(const VhAttnArgs& p, int q, int key, int kend) {
    if (key >= kend) return false;
    if (p.causal && key > q + p.q_off) return false;
    if (p.chunk > 0) {
        const int ci = q / p.chunk;
        const int start = (p.left < 0) ? 0 : max((ci - p.left) * p.chunk, 0);
        const int end = min((ci + 1) * p.chunk, p.Sk);
        if (key < start || key >= end) return false;
    }
    return true;
}

template <int D, bool REL>
__global__ __launch_bounds__(256) void k_attn(const VhAttnArgs p) {
    constexpr int KSTR = D + 2, VSTR = D + 16, NS = D / 4, NT = D / 16;
    __shared__ __attribute__((aligned(16))) float Kt[AT_KT * KSTR];
    __shared__ __attribute__((aligned(16))) float Vt[AT_KT * VSTR];
    __shared__ __attribute__((aligned(16))) float Pt[REL ? AT_KT * KSTR : 2];
    __shared__ __attribute__((aligned(16))) float Ps[4][16 * AT_PSTR];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int hk = h / (p.Hq / p.Hkv);
    const int qblk = blockIdx.x * 64;
    const int q0 = qblk + wid * 16;
    const int lr = lane & 15, lg = lane >> 4;

    const float* Qb = p.Q + (size_t)b * p.bsq + (size_t)h * p.hsq;
    const float* Kb = p.K + (size_t)b * p.bsk + (size_t)hk * p.hsk;
    const float* Vb = p.V + (size_t)b * p.bsk + (size_t)hk * p.hsv;
    const float* Pb = REL ? (p.P + (size_t)h * p.hsp) : nullptr;

    // Q fragments (A operand): a_s = Q[q0 + lr][4s + lg]
    float qa[NS];
    float qb[REL ? NS : 1];
    {
        const int q = q0 + lr;
        const bool ok = q < p.Sq;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int dd = 4 * s + lg;
            const float v = ok ? Qb[(size_t)q * p.ldq + dd] : 0.f;
            if (REL) {
                qa[s] = v + p.bias_u[h * D + dd];
                qb[s] = v + p.bias_v[h * D + dd];
            } else {
                qa[s] = v;
            }
        }
    }

    int kend = min(p.Sk, p.klen);
    int kloop = kend;
    if (p.causal) kloop = min(kloop, min(qblk + 63, p.Sq - 1) + p.q_off + 1);

    float m[4], l[4];
    f32x4 o[NT];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
#pragma unroll
    for (int t = 0; t < NT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K/V(/P) tiles are prefetched one tile ahead into NATIVE vector registers with unconditional, clamped
    // 16-byte loads (rows past Sk are zeroed when the tile is written to LDS): the loads of tile t+1 are in
    // flight during the MFMAs of tile t.  (The first version loaded 8 bytes at a time under a branch inside
    // the staging loop: every tile began with a synchronous global round trip.)
    constexpr int F4 = AT_KT * (D / 4) / 256;   // 16-byte pieces per thread per operand: 2 (D=64) / 4 (D=128)
    f32x4 kr[F4], vr[F4], pr[REL ? F4 : 1];
    auto load_tile = [&](int kt0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / (D / 4), c4 = idx % (D / 4);
            int key = min(kt0 + row, p.Sk - 1);
            if (p.ktable) key = p.ktable[key >> 6] * 64 + (key & 63);       // paged KV cache (Mixtral prefill of a sequence)
            kr[i] = reinterpret_cast<const f32x4*>(Kb + (size_t)key * p.ldk)[c4];
            vr[i] = reinterpret_cast<const f32x4*>(Vb + (size_t)key * p.ldv)[c4];
            if (REL) pr[i] = reinterpret_cast<const f32x4*>(Pb + (size_t)key * p.ldp)[c4];
        }
    };
    auto store_tile = [&](int kt0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / (D / 4), c4 = idx % (D / 4);
            const bool ok = kt0 + row < p.Sk;
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 kv = ok ? kr[i] : z, vv = ok ? vr[i] : z;
            float* kd = &Kt[row * KSTR + c4 * 4];          // KSTR = D + 2: 8-byte aligned rows
            *reinterpret_cast<float2*>(kd) = make_float2(kv[0], kv[1]);
            *reinterpret_cast<float2*>(kd + 2) = make_float2(kv[2], kv[3]);
            *reinterpret_cast<f32x4*>(&Vt[row * VSTR + c4 * 4]) = vv;   // VSTR = D + 16: 16-byte aligned rows
            if (REL) {
                const f32x4 pv = ok ? pr[i] : z;
                float* pd = &Pt[row * KSTR + c4 * 4];
                *reinterpret_cast<float2*>(pd) = make_float2(pv[0], pv[1]);
                *reinterpret_cast<float2*>(pd + 2) = make_float2(pv[2], pv[3]);
            }
        }
    };

    if (kloop > 0) load_tile(0);
    for (int kt0 = 0; kt0 < kloop; kt0 += AT_KT) {
        store_tile(kt0);
        __syncthreads();
        load_tile(min(kt0 + AT_KT, max(kloop - 1, 0)));   // always issued (clamped): counted by the compiler

        // ---- S = Q K^T (+ Qv P^T) for two 16-key sub-tiles ---------------------------
        f32x4 sacc[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* kr = &Kt[(jt * 16 + lr) * KSTR + lg];
#pragma unroll
            for (int s = 0; s < NS; ++s) a = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s], kr[4 * s], a, 0, 0, 0);
            if (REL) {
                const float* pr = &Pt[(jt * 16 + lr) * KSTR + lg];
#pragma unroll
                for (int s = 0; s < NS; ++s) a = __builtin_amdgcn_mfma_f32_16x16x4f32(qb[s], pr[4 * s], a, 0, 0, 0);
            }
            sacc[jt] = a;
        }

        // ---- online softmax in D layout --------------------------------------------
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = q0 + lg * 4 + r;
            float s0 = sacc[0][r] * p.scale, s1 = sacc[1][r] * p.scale;
            if (!key_visible(p, q, kt0 + lr, kend)) s0 = -INFINITY;
            if (!key_visible(p, q, kt0 + 16 + lr, kend)) s1 = -INFINITY;
            const float mx = grp16_max(fmaxf(s0, s1));
            const float mn = fmaxf(m[r], mx);
            float p0, p1;
            if (mn == -INFINITY) {
                alpha[r] = 1.f; p0 = 0.f; p1 = 0.f;
            } else {
                alpha[r] = __expf(m[r] - mn);
                p0 = __expf(s0 - mn);
                p1 = __expf(s1 - mn);
            }
            l[r] = l[r] * alpha[r] + grp16_sum(p0 + p1);
            m[r] = mn;
            Ps[wid][(lg * 4 + r) * AT_PSTR + lr] = p0;
            Ps[wid][(lg * 4 + r) * AT_PSTR + 16 + lr] = p1;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[t][r] *= alpha[r];
        __syncthreads();  // P patch visible to the whole wave (block-uniform trip count)

        // ---- O += P V --------------------------------------------------------------
#pragma unroll
        for (int s = 0; s < AT_KT / 4; ++s) {
            const float pa = Ps[wid][lr * AT_PSTR + 4 * s + lg];
            const float* vr = &Vt[(4 * s + lg) * VSTR + lr];
#pragma unroll
            for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, vr[16 * t], o[t], 0, 0, 0);
        }
        __syncthreads();  // tiles and P patch are rewritten next iteration
    }

    float* Ob = p.O + (size_t)b * p.bso + (size_t)h * D;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + lg * 4 + r;
        if (q >= p.Sq) continue;
        const float inv = (l[r] > 0.f) ? 1.0f / l[r] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) Ob[(size_t)q * p.ldo + 16 * t + lr] = o[t][r] * inv;
    }
}

}  // namespace

int vhk_attn(hipStream_t st, const VhAttnArgs& a) {
    if (a.Sq <= 0 || a.Sk <= 0 || a.Hq % a.Hkv != 0) return -1;
    const dim3 grid((a.Sq + 63) / 64, a.Hq, a.B), blk(256);
    const bool rel = a.P != nullptr;
    if (a.d == 64 && !rel) hipLaunchKernelGGL((k_attn<64, false>), grid, blk, 0, st, a);
    else if (a.d == 64 && rel) hipLaunchKernelGGL((k_attn<64, true>), grid, blk, 0, st, a);
    else if (a.d == 128 && !rel) hipLaunchKernelGGL((k_attn<128, false>), grid, blk, 0, st, a);
    else return -1;
    return 0;
}
