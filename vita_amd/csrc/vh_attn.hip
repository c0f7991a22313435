// vh_attn.hip — multi-query-row attention on the matrix cores, operands straight from L2 (no LDS tiles).
// Serves three call sites:
//   K5  InternViT global attention, 16 heads x 64, N=1025, no mask
//       (internvit/modeling_intern_vit.py:158-177)                                   -> k_attn_x3 (bf16 x 3 MFMAs)
//   K14 Whale rel-pos attention, 16 heads x 64: scores = ((q+u)k^T + (q+v)p^T)/sqrt(d),
//       NO rel-shift, pad/chunk mask -> excluded keys (whale/module/layer/attention.py:370-419,
//       whale/utils.py:88-146)                                                       -> k_attn_direct (fp32 MFMA)
//   K23 Mixtral prefill: causal GQA, 32 q-heads / 8 kv-heads x 128, K/V read from the fp32
//       KV cache (HF eager_attention_forward, modeling_mixtral.py:256-279)           -> k_attn_x3
// One wave owns 16 (or 32) query rows and a share of the keys; flash-style online softmax in the MFMA D layout
// (row = (lane>>4)*4 + r, col = lane&15) with DPP row reductions; probabilities go D-layout -> A-layout through a
// wave-private LDS patch.  (r01's LDS-tiled 4-wave kernel was removed in r04: 143 us on the ViT against 50 here.)
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


// ---- block -> (q tile, head, batch), XCD-aware (r06) -----------------------------------------------------------------------------------
// Hardware deals the blocks of a grid to the 8 XCDs round-robin in linear order (x fastest), and every XCD has its own 4 MB L2.  With the
// plain decode (x = q tile, y = head, z = batch) the q tiles of ONE head — which all stream the same K / V — land on all 8 XCDs and each L2
// fetches every head's K / V: at S = 2344 the flash kernel's FETCH_SIZE was 190 MB per launch for 19 MB of tile images + 38 MB of Q (65 MB now).
// xcd_map: linear block L of a chunk of 8 (head, batch) pairs takes pair L % 8 and q tile L / 8: all q tiles of a pair run on ONE XCD,
// whose L2 then holds that pair's K / V only (2.4 MB of images per KV head at S = 2344).  Bijective for any grid (a last chunk of r < 8
// pairs deals L % r); q tiles stay in dispatch order.
__device__ __forceinline__ void at_block_coords(const int xcd_map, int& qx, int& hy, int& bz) {
    qx = blockIdx.x; hy = blockIdx.y; bz = blockIdx.z;
    if (!xcd_map) return;
    const int nx = gridDim.x, ny = gridDim.y, np = ny * (int)gridDim.z;
    const int L = qx + nx * (hy + ny * bz);
    const int chunk = L / (8 * nx), base = chunk * 8;
    const int r = min(8, np - base);
    const int Lc = L - chunk * 8 * nx;
    const int pair = base + Lc % r;
    qx = Lc / r;
    hy = pair % ny;
    bz = pair / ny;
}

// ---- direct-operand variant (default) ----------------------------------------------------------------------------
// One wave = 16 query rows x one share of the keys; NO LDS tiles and no block barriers in the loop.  The MFMA operand
// layout of v_mfma_f32_16x16x4_f32 lets every lane fetch its B operands as 16-byte global loads when the reduction
// index is PERMUTED consistently on both operands:
//   S = Q K^T : the four MFMAs of chunk c use d = 16c + 4*lg + u (u = 0..3): lane (lr, lg) holds the float4
//               K[key lr][16c + 4lg ..] and the float4 Q[row lr][16c + 4lg ..] — a dot product does not care about order;
//   O += P V  : lane lr owns the output columns d = lr*(D/16) + t, so its B operand of step s is the float4 (two for
//               d = 128) V[key 4s + lg][lr*(D/16) ..]: 4 keys x a whole row per wave instruction, fully coalesced, and the
//               result is stored with 16-byte stores.
// Operands come straight from L2/L1 (K/V of a head: 0.5 MB, shared by every query tile of the head and, under GQA, by 4
// heads), P goes D-layout -> A-layout through a wave-private LDS patch (wave-level ordering only).  The K registers are
// reloaded for the next tile as soon as S is formed, the V registers after the PV product, so every load has a whole
// phase to land.  KS waves of a block deal the key tiles among themselves (tile t -> wave t % KS) and merge (m, l, O) once
// at the end — the only block barrier.  Waves never wait for each other, so occupancy is whatever the registers allow
// and the launch is 16-row tiles x heads blocks (ViT: 1040) instead of 272 four-wave blocks marching through barriers.
template <int D, bool REL, int KS, int WPE>
__global__ __launch_bounds__(64 * KS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_attn_direct(const VhAttnArgs p) {
    constexpr int NC = D / 16;          // 16-wide chunks of d = K float4s per key row per lane-group = accumulators
    constexpr int VQ = NC / 4;          // float4s of V per lane per key
    constexpr int MGW = 8 + NC * 4;
    __shared__ __attribute__((aligned(16))) float Ps[KS][16 * AT_PSTR];
    __shared__ __attribute__((aligned(16))) float Mg[(KS > 1 ? KS - 1 : 1) * 64 * MGW];

    const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6;
    int qx, h, b;
    at_block_coords(p.xcd_map, qx, h, b);
    const int hk = h / (p.Hq / p.Hkv);
    const int q0 = qx * 16;
    const int lr = lane & 15, lg = lane >> 4;
    float* ps = Ps[kg];

    const float* Qb = p.Q + (size_t)b * p.bsq + (size_t)h * p.hsq;
    const float* Kb = p.K + (size_t)b * p.bsk + (size_t)hk * p.hsk;
    const float* Vb = p.V + (size_t)b * p.bsk + (size_t)hk * p.hsv;
    const float* Pb = REL ? (p.P + (size_t)h * p.hsp) : nullptr;

    f32x4 qa[NC], qb[REL ? NC : 1];
    {
        const int q = min(q0 + lr, p.Sq - 1);               // rows past Sq compute on a copy of the last row, never stored
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(Qb + (size_t)q * p.ldq + 16 * c + 4 * lg);
            if (REL) {
                qa[c] = v + *reinterpret_cast<const f32x4*>(p.bias_u + h * D + 16 * c + 4 * lg);
                qb[c] = v + *reinterpret_cast<const f32x4*>(p.bias_v + h * D + 16 * c + 4 * lg);
            } else {
                qa[c] = v;
            }
        }
    }

    const int kend = min(p.Sk, p.klen);
    int kloop = kend;
    if (p.causal) kloop = min(kloop, min(q0 + 15, p.Sq - 1) + p.q_off + 1);

    float m[4], l[4];
    f32x4 o[NC];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
#pragma unroll
    for (int t = 0; t < NC; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 kf[2][NC], pf[REL ? 2 : 1][REL ? NC : 1], vf[AT_KT / 4][VQ];
    auto phys = [&](int key) __attribute__((always_inline)) {
        key = min(key, max(kend, 1) - 1);        // masked keys re-read the last VISIBLE row (rows in [klen, Sk) may hold anything)
        return p.ktable ? p.ktable[key >> 6] * 64 + (key & 63) : key;
    };
    auto load_k = [&](int kt0) __attribute__((always_inline)) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const size_t row = (size_t)phys(kt0 + 16 * jt + lr);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                kf[jt][c] = *reinterpret_cast<const f32x4*>(Kb + row * p.ldk + 16 * c + 4 * lg);
                if (REL) pf[jt][c] = *reinterpret_cast<const f32x4*>(Pb + row * p.ldp + 16 * c + 4 * lg);
            }
        }
    };
    auto load_v = [&](int kt0) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < AT_KT / 4; ++s) {
            const size_t row = (size_t)phys(kt0 + 4 * s + lg);
#pragma unroll
            for (int j = 0; j < VQ; ++j) vf[s][j] = *reinterpret_cast<const f32x4*>(Vb + row * p.ldv + lr * NC + 4 * j);
        }
    };

    constexpr int STEP = KS * AT_KT;
    int kt0 = kg * AT_KT;
    load_k(kt0);
    load_v(kt0);
    for (; kt0 < kloop; kt0 += STEP) {
        // ---- S = Q K^T (+ Qv P^T) for the two 16-key sub-tiles ------------------------------------------------------
        f32x4 sacc[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int u = 0; u < 4; ++u) a = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[c][u], kf[jt][c][u], a, 0, 0, 0);
            if (REL) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int u = 0; u < 4; ++u) a = __builtin_amdgcn_mfma_f32_16x16x4f32(qb[c][u], pf[jt][c][u], a, 0, 0, 0);
            }
            sacc[jt] = a;
        }
        load_k(kt0 + STEP);                                  // (clamped) K of this wave's next tile: lands under softmax + PV

        // ---- online softmax in D layout: lane holds S[q0 + 4lg + r][kt0 + 16jt + lr] --------------------------------
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
            ps[(lg * 4 + r) * AT_PSTR + lr] = p0;
            ps[(lg * 4 + r) * AT_PSTR + 16 + lr] = p1;
        }
#pragma unroll
        for (int t = 0; t < NC; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[t][r] *= alpha[r];
        // the patch is private to this wave: its LDS operations execute in order, only the compiler must not reorder them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- O += P V -------------------------------------------------------------------------------------------
#pragma unroll
        for (int s = 0; s < AT_KT / 4; ++s) {
            const float pa = ps[lr * AT_PSTR + 4 * s + lg];
#pragma unroll
            for (int t = 0; t < NC; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, vf[s][t >> 2][t & 3], o[t], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // patch reads are done before the next tile rewrites it
        load_v(kt0 + STEP);                                  // lands under the next S and softmax
    }

    // ---- merge the key shares into wave 0, in wave order (deterministic) ---------------------------------------------
    if (KS > 1) {
        if (kg > 0) {
            float* mg = Mg + ((kg - 1) * 64 + lane) * MGW;
#pragma unroll
            for (int r = 0; r < 4; ++r) { mg[r] = m[r]; mg[4 + r] = l[r]; }
#pragma unroll
            for (int t = 0; t < NC; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mg[8 + t * 4 + r] = o[t][r];
        }
        __syncthreads();
        if (kg > 0) return;
        for (int g = 1; g < KS; ++g) {
            const float* mg = Mg + ((g - 1) * 64 + lane) * MGW;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m2 = mg[r], l2 = mg[4 + r];
                const float mn = fmaxf(m[r], m2);
                const float a1 = (mn == -INFINITY) ? 1.f : __expf(m[r] - mn);
                const float a2 = (mn == -INFINITY) ? 0.f : __expf(m2 - mn);
                l[r] = l[r] * a1 + l2 * a2;
                m[r] = mn;
#pragma unroll
                for (int t = 0; t < NC; ++t) o[t][r] = o[t][r] * a1 + mg[8 + t * 4 + r] * a2;
            }
        }
    }

    float* Ob = p.O ? p.O + (size_t)b * p.bso + (size_t)h * D : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + lg * 4 + r;
        if (q >= p.Sq) continue;
        const float inv = (l[r] > 0.f) ? 1.0f / l[r] : 0.f;
#pragma unroll
        for (int j = 0; j < VQ; ++j) {
            const f32x4 v = f32x4{o[4 * j][r] * inv, o[4 * j + 1][r] * inv, o[4 * j + 2][r] * inv, o[4 * j + 3][r] * inv};
            if (Ob) *reinterpret_cast<f32x4*>(Ob + (size_t)q * p.ldo + lr * NC + 4 * j) = v;
            if (p.O_hi) {   // (r03) the O projection streams bf16 hi/lo planes: emitted here instead of by a split pass
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) split_bf16(v[i], hi[i], lo[i]);
                const size_t at = ((size_t)b * p.Sq + q) * p.ldo_split + (size_t)h * D + lr * NC + 4 * j;
                *reinterpret_cast<uint2*>(p.O_hi + at) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
                *reinterpret_cast<uint2*>(p.O_lo + at) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
            }
        }
    }
}


// ---- bf16 x 3 variant of the direct-operand kernel (r03; default for the plain and causal call sites) -----------------
// Same wave = 16 query rows x a share of the keys, same online softmax in the D layout, same merge — but the two matrix
// products run on v_mfma_f32_16x16x32_bf16 with BOTH operands split into bf16 hi + lo (x = hi + lo to 2^-17, hardware
// v_cvt_pk_bf16_f32) and three products per fragment pair (lo*hi + hi*lo + hi*hi; lo*lo is below 2^-17 of the term): the
// precision class of the exact-mode GEMMs (vh_gemm.hip) at 3/16 of the matrix-pipe time of the fp32 MFMA
// (16x16x4 f32: 2048 FLOP in 32 cycles; 16x16x32 bf16: 16384 FLOP in 16).  The 32-deep reduction changes the operand
// shapes, not the loads:
//   S = Q K^T : lane (lr, lg) holds Q[row lr][32c + 8lg ..+8] and K[key lr][32c + 8lg ..+8] — two float4 each per chunk c;
//   O += P V  : A = P[row lr][keys 8lg ..+8] (8 consecutive floats of the wave-private LDS patch, one ds_read_b128 pair),
//               B_t = V[keys 8lg + u][lr*(D/16) + t], u = 0..7: the lane loads V[key][lr*(D/16) ..] for its 8 keys (as the
//               fp32 kernel does for 4) and pairs the values ALONG THE KEYS when it converts them (v_cvt_pk takes two
//               registers), so the transposition costs nothing.
// Rel-pos attention (Whale, 12 us) stays on the fp32 kernel.
typedef __attribute__((ext_vector_type(2))) float at_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 at_bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned int at_u32x4;
__device__ __forceinline__ void at_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const at_f32x2 v = {a, b};
    const at_bf16x2 h = __builtin_convertvector(v, at_bf16x2);
    const at_f32x2 r = v - __builtin_convertvector(h, at_f32x2);
    const at_bf16x2 l = __builtin_convertvector(r, at_bf16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ void at_split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
    uint32_t h[4], l[4];
    at_split2(a[0], a[1], h[0], l[0]);
    at_split2(a[2], a[3], h[1], l[1]);
    at_split2(b[0], b[1], h[2], l[2]);
    at_split2(b[2], b[3], h[3], l[3]);
    hi = __builtin_bit_cast(bf16x8, at_u32x4{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(bf16x8, at_u32x4{l[0], l[1], l[2], l[3]});
}
#define AT_PSTR3 36     // patch row stride (floats): 16-byte aligned 8-float reads at 8 lg, rows 4 banks apart mod 64

// RT = 16-row tiles per wave.  With one tile the launch is bound by operand traffic, not by either pipe: every wave pulls
// the whole K and V of its head through L2 -> registers (ViT: 4160 waves x 128 KB = 532 MB per launch, 8.7 TB/s at 61 us;
// the fp32 kernel moved the same bytes in 78 us).  Two tiles per wave (d = 64) halve the bytes and the conversions per row.
// MODE fixes the mask flavour at compile time (0 = pad mask only, 1 = causal, 2 = causal + KV page table): with the flavours
// as run-time branches the loop body held 34 branches / 20 exec-mask regions, each a scheduling barrier between the loads,
// conversions and MFMAs it should interleave.
// (r03's pre-pass that converted K / V to bf16 planes once per launch — 160 of the loop's 510 instructions are conversions — lost to
// its own launch: ViT 66 vs 61 us, prefill 78 vs 49 us; removed in r04.  The flash form below converts once per BLOCK instead.)
template <int D, int KS, int WPE, int RT, int MODE>
__global__ __launch_bounds__(64 * KS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_attn_x3(const VhAttnArgs p) {
    constexpr bool CAUSAL = MODE >= 1, PAGED = MODE == 2;
    constexpr int NC = D / 16;          // output column tiles = floats of a V row per lane
    constexpr int VQ = NC / 4;          // float4s of V per lane per key
    constexpr int C32 = D / 32;         // 32-deep chunks of the head dimension
    constexpr int MGW = RT * (8 + NC * 4);
    __shared__ __attribute__((aligned(16))) float Ps[KS][RT][16 * AT_PSTR3];
    __shared__ __attribute__((aligned(16))) float Mg[(KS > 1 ? KS - 1 : 1) * 64 * MGW];

    const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6;
    int qx, h, b;
    at_block_coords(p.xcd_map, qx, h, b);
    const int hk = h / (p.Hq / p.Hkv);
    const int q0 = qx * 16 * RT;
    const int lr = lane & 15, lg = lane >> 4;

    const float* Qb = p.Q + (size_t)b * p.bsq + (size_t)h * p.hsq;
    const float* Kb = p.K + (size_t)b * p.bsk + (size_t)hk * p.hsk;
    const float* Vb = p.V + (size_t)b * p.bsk + (size_t)hk * p.hsv;

    const float qscale = p.scale * 1.44269504088896340736f;
    bf16x8 qh[RT][C32], ql[RT][C32];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int q = min(q0 + 16 * rt + lr, p.Sq - 1);     // rows past Sq compute on a copy of the last row, never stored
#pragma unroll
        for (int c = 0; c < C32; ++c) {
            // scale * log2(e) folded into Q: the scores come out of the MFMAs in the log2 domain, softmax runs on v_exp_f32 directly
            const float* src = Qb + (size_t)q * p.ldq + 32 * c + 8 * lg;
            at_split8(*reinterpret_cast<const f32x4*>(src) * qscale, *reinterpret_cast<const f32x4*>(src + 4) * qscale, qh[rt][c], ql[rt][c]);
        }
    }

    const int kend = min(p.Sk, p.klen);
    int kloop = kend;
    if (CAUSAL) kloop = min(kloop, min(q0 + 16 * RT - 1, p.Sq - 1) + p.q_off + 1);

    float m[RT][4], l[RT][4];
    f32x4 o[RT][NC];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { m[rt][r] = -INFINITY; l[rt][r] = 0.f; }
#pragma unroll
        for (int t = 0; t < NC; ++t) o[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    f32x4 kf[2][C32][2], vf[8][VQ];
    // Row addresses are a uniform base + a 32-bit byte offset from a 24-bit multiply (the launcher checks that rows x stride
    // fits): the first version spent 63 quarter-rate 32/64-bit multiplies and ~100 more VALU per tile on 64-bit row pointers,
    // a third of the loop's issue cycles in a kernel that is VALU-bound.
    const unsigned ldk = (unsigned)p.ldk, ldv = (unsigned)p.ldv;
    const int klast = max(kend, 1) - 1;   // masked keys re-read the last VISIBLE row: rows in [klen, Sk) are caller memory (NaN bits there would turn p = 0 into NaN in the PV products)
    auto row_bytes = [&](int key, unsigned ld, unsigned col) __attribute__((always_inline)) {
        key = min(key, klast);
        if (PAGED) key = p.ktable[key >> 6] * 64 + (key & 63);
        return (__umul24((unsigned)key, ld) + col) * 4u;
    };
    auto load_k = [&](int kt0) __attribute__((always_inline)) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const char* row = reinterpret_cast<const char*>(Kb) + row_bytes(kt0 + 16 * jt + lr, ldk, 8 * lg);
#pragma unroll
            for (int c = 0; c < C32; ++c) {
                kf[jt][c][0] = *reinterpret_cast<const f32x4*>(row + 128 * c);
                kf[jt][c][1] = *reinterpret_cast<const f32x4*>(row + 128 * c + 16);
            }
        }
    };
    auto load_v = [&](int kt0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const char* row = reinterpret_cast<const char*>(Vb) + row_bytes(kt0 + 8 * lg + u, ldv, lr * NC);
#pragma unroll
            for (int j = 0; j < VQ; ++j) vf[u][j] = *reinterpret_cast<const f32x4*>(row + 16 * j);
        }
    };

    constexpr int STEP = KS * AT_KT;
    int kt0 = kg * AT_KT;
    load_k(kt0);
    load_v(kt0);
    for (; kt0 < kloop; kt0 += STEP) {
        // ---- S = Q K^T for the two 16-key sub-tiles: 3 bf16 products per 32-deep chunk, K converted once for all row tiles
        f32x4 sacc[RT][2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) sacc[rt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < C32; ++c) {
                bf16x8 kh, kl;
                at_split8(kf[jt][c][0], kf[jt][c][1], kh, kl);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x4 a = sacc[rt][jt];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ql[rt][c], kh, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[rt][c], kl, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[rt][c], kh, a, 0, 0, 0);
                    sacc[rt][jt] = a;
                }
            }
        }
        load_k(kt0 + STEP);                                  // (clamped) K of this wave's next tile: lands under softmax + PV

        // ---- online softmax in D layout: lane holds S[q0 + 16rt + 4lg + r][kt0 + 16jt + lr] ---------------------------
        // every key of the tile visible to every row of the wave (first row's causal limit, pad limit)?
        const bool full_tile = kt0 + AT_KT <= (CAUSAL ? min(kend, q0 + p.q_off + 1) : kend);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float* ps = Ps[kg][rt];
            float alpha[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + 16 * rt + lg * 4 + r;
                float s0 = sacc[rt][0][r], s1 = sacc[rt][1][r];
                if (!full_tile) {                                                 // (wave-uniform) only the tiles that cross a mask edge
                    const int klim = CAUSAL ? min(kend, q + p.q_off + 1) : kend;  // keys < klim are visible
                    s0 = (kt0 + lr < klim) ? s0 : -INFINITY;
                    s1 = (kt0 + 16 + lr < klim) ? s1 : -INFINITY;
                }
                const float mx = grp16_max(fmaxf(s0, s1));
                const float mn = fmaxf(m[rt][r], mx);
                const bool none = mn == -INFINITY;                                // nothing visible yet: exp2(-inf + inf) is discarded
                alpha[r] = none ? 1.f : __builtin_amdgcn_exp2f(m[rt][r] - mn);
                const float p0 = none ? 0.f : __builtin_amdgcn_exp2f(s0 - mn);
                const float p1 = none ? 0.f : __builtin_amdgcn_exp2f(s1 - mn);
                l[rt][r] = l[rt][r] * alpha[r] + grp16_sum(p0 + p1);
                m[rt][r] = mn;
                ps[(lg * 4 + r) * AT_PSTR3 + lr] = p0;
                ps[(lg * 4 + r) * AT_PSTR3 + 16 + lr] = p1;
            }
#pragma unroll
            for (int t = 0; t < NC; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[rt][t][r] *= alpha[r];
        }
        // the patches are private to this wave: its LDS operations execute in order, only the compiler must not reorder them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- O += P V: one 32-key step, 3 products per output column tile, V converted once for all row tiles ----------
        bf16x8 ph[RT], pl[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float* ps = Ps[kg][rt];
            at_split8(*reinterpret_cast<const f32x4*>(ps + lr * AT_PSTR3 + 8 * lg),
                      *reinterpret_cast<const f32x4*>(ps + lr * AT_PSTR3 + 8 * lg + 4), ph[rt], pl[rt]);
        }
#pragma unroll
        for (int t = 0; t < NC; ++t) {
            bf16x8 bh, bl;
            {
                uint32_t vh[4], vl[4];
#pragma unroll
                for (int u2 = 0; u2 < 4; ++u2)
                    at_split2(vf[2 * u2][t >> 2][t & 3], vf[2 * u2 + 1][t >> 2][t & 3], vh[u2], vl[u2]);
                bh = __builtin_bit_cast(bf16x8, at_u32x4{vh[0], vh[1], vh[2], vh[3]});
                bl = __builtin_bit_cast(bf16x8, at_u32x4{vl[0], vl[1], vl[2], vl[3]});
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                o[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[rt], bh, o[rt][t], 0, 0, 0);
                o[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[rt], bl, o[rt][t], 0, 0, 0);
                o[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[rt], bh, o[rt][t], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // patch reads are done before the next tile rewrites them
        load_v(kt0 + STEP);                                  // lands under the next S and softmax
    }

    // ---- merge the key shares into wave 0, in wave order (deterministic) ---------------------------------------------
    if (KS > 1) {
        if (kg > 0) {
            float* mg = Mg + ((kg - 1) * 64 + lane) * MGW;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float* mr = mg + rt * (8 + NC * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { mr[r] = m[rt][r]; mr[4 + r] = l[rt][r]; }
#pragma unroll
                for (int t = 0; t < NC; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mr[8 + t * 4 + r] = o[rt][t][r];
            }
        }
        __syncthreads();
        if (kg > 0) return;
        for (int g = 1; g < KS; ++g) {
            const float* mg = Mg + ((g - 1) * 64 + lane) * MGW;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float* mr = mg + rt * (8 + NC * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m2 = mr[r], l2 = mr[4 + r];
                    const float mn = fmaxf(m[rt][r], m2);
                    const float a1 = (mn == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m[rt][r] - mn);
                    const float a2 = (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
                    l[rt][r] = l[rt][r] * a1 + l2 * a2;
                    m[rt][r] = mn;
#pragma unroll
                    for (int t = 0; t < NC; ++t) o[rt][t][r] = o[rt][t][r] * a1 + mr[8 + t * 4 + r] * a2;
                }
            }
        }
    }

    float* Ob = p.O ? p.O + (size_t)b * p.bso + (size_t)h * D : nullptr;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = q0 + 16 * rt + lg * 4 + r;
            if (q >= p.Sq) continue;
            const float inv = (l[rt][r] > 0.f) ? 1.0f / l[rt][r] : 0.f;
#pragma unroll
            for (int j = 0; j < VQ; ++j) {
                const f32x4 v = f32x4{o[rt][4 * j][r] * inv, o[rt][4 * j + 1][r] * inv, o[rt][4 * j + 2][r] * inv, o[rt][4 * j + 3][r] * inv};
                if (Ob) *reinterpret_cast<f32x4*>(Ob + (size_t)q * p.ldo + lr * NC + 4 * j) = v;
                if (p.O_hi) {
                    uint32_t hi[2], lo[2];
                    at_split2(v[0], v[1], hi[0], lo[0]);
                    at_split2(v[2], v[3], hi[1], lo[1]);
                    const size_t at = ((size_t)b * p.Sq + q) * p.ldo_split + (size_t)h * D + lr * NC + 4 * j;
                    *reinterpret_cast<uint2*>(p.O_hi + at) = make_uint2(hi[0], hi[1]);
                    *reinterpret_cast<uint2*>(p.O_lo + at) = make_uint2(lo[0], lo[1]);
                }
            }
        }
    }
}


// ---- flash form of the bf16 x 3 kernel for the Mixtral prefill (r04): K / V tiles shared by a block's waves through LDS ------------
// The direct kernel above has every wave pull its keys' K and V rows from L2 in FRAGMENT shape (16 rows x 32 B per wave
// instruction) and convert them to bf16 hi/lo itself; at d = 128 that is 308 MB through the vector-memory path per S = 552
// launch (5.6 GB at S = 2344) at ~6 TB/s, which is what the launch lasts.  Here a block is ONE KV head x 16 query rows x its FOUR
// query heads (Mixtral's 4 : 1 grouping), EIGHT waves = the four heads x the two 32-key halves of every 64-key tile, and every
// tile of K and V is read from global ONCE per block in whole rows (K by wave group 1: two 512-byte rows per wave instruction;
// V by group 0), converted ONCE by the thread that loaded it, and laid in LDS as MFMA-ready bf16 hi/lo images:
//   K planes [half][64 keys][128 B], 16-byte chunks XOR-swizzled by (key >> 1) & 7: the B fragment of S = Q K^T (key lr,
//            chunk 4c + lg) is one conflict-free ds_read_b128 (the layout of vh_gemm.hip's operand tiles);
//   V planes TRANSPOSED [rho(col)][64 keys] with rho(d) = 16 (d % 8) + d / 8: row 16 t + lr holds column 8 lr + t, so the B
//            fragment of O += P V (keys 32 s + 8 lg ..+8 of the lane's column of output tile t) is one ds_read_b128 and the output
//            keeps the direct kernel's lane -> 8 consecutive columns mapping (16-byte stores).  The transposition is free:
//            v_cvt_pk pairs two registers, here the same column of two consecutive keys.
// Two LDS buffers, register staging one tile ahead: the loads of tile t + 2 are issued behind the barrier that ends tile t, the
// conversion + LDS writes of tile t + 1 run at the end of tile t into the buffer last read in tile t - 1: ONE barrier per 64 keys.
// Wave group kg takes half kg of every tile; the groups merge (m, l, O) once at the end, like the direct kernel's key groups.
// (With ONE group — 4 waves, both halves in turn — every dependent latency of a half tile is exposed at one wave per SIMD: 42 vs
// 34 us at S = 552.)  The arithmetic per 32-key half tile (products, softmax, accumulation order) is the direct kernel's.
// Measured (profiles/r04_attn_fa_v2.jsonl): S = 552 46.8 -> 34.3 us, S = 2344 468 -> 310 us.  A d = 64 instantiation for the ViT
// (block = one head x 64 / 128 rows) was built and measured too: 50-63 vs 47 us on one image, a tie on eight — not kept.
// IMG (r06): the tiles arrive as the producer's MFMA-ready images (VhAttnArgs::kv_img, written by k_rope_kv_img in this kernel's own
// LDS layout): a tile is 8 LDS-DMA pieces of 16 B per thread straight into the buffer (global_load_lds: no staging registers, no
// conversion, no ds_write — a third of the IMG = false kernel, profiles/r04_attn_fa_ablate.txt), issued for tile t + 1 at the head of
// tile t into the buffer everyone left at the barrier before, waited for (counted vmcnt, inline asm: hipcc does not see these loads)
// in front of the barrier that ends tile t.
__device__ __forceinline__ void fa_glds16(const unsigned char* base, uint32_t off, unsigned char* lds_dst) {
    const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_dst;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
}
template <int MODE, bool IMG, int RT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_attn_fa(const VhAttnArgs p) {
    constexpr bool CAUSAL = MODE >= 1, PAGED = MODE == 2;
    constexpr int D = 128, NC = D / 16, VQ = NC / 4, C32 = D / 32;
    constexpr int PL = 64 * D * 2;                 // bytes of one plane of one 64-key tile
    constexpr int MGW = 8 + NC * 4;
    static_assert(4 * 64 * RT * MGW * 4 <= 8 * PL, "merge area aliases the K / V buffers");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[8 * PL + 8 * 16 * AT_PSTR3 * 4];
    float* patches = reinterpret_cast<float*>(lds + 8 * PL);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rg = w & 3, kg = w >> 2;                        // head of the KV group; key half
    const int ftid = tid & 255;                               // index inside the staging role
    const int lr = lane & 15, lg = lane >> 4;
    int qx, hk, b;
    at_block_coords(p.xcd_map, qx, hk, b);                    // (y = KV head here: the blocks of a KV head share its tile images)
    const int qb = (int)gridDim.x - 1 - qx;                   // heaviest (latest rows under the causal mask) blocks first
    const int h = hk * 4 + rg, q0 = qb * 16 * RT;
    const int qlast = min(q0 + 16 * RT, p.Sq) - 1;

    const float* Qb = p.Q + (size_t)b * p.bsq + (size_t)h * p.hsq;
    const float* Kb = p.K + (size_t)b * p.bsk + (size_t)hk * p.hsk;
    const float* Vb = p.V + (size_t)b * p.bsk + (size_t)hk * p.hsv;

    const float qscale = p.scale * 1.44269504088896340736f;
    bf16x8 qh[RT][C32], ql[RT][C32];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int q = min(q0 + 16 * rt + lr, p.Sq - 1);        // rows past Sq compute on a copy of the last row, never stored
#pragma unroll
        for (int c = 0; c < C32; ++c) {
            const float* src = Qb + (size_t)q * p.ldq + 32 * c + 8 * lg;
            at_split8(*reinterpret_cast<const f32x4*>(src) * qscale, *reinterpret_cast<const f32x4*>(src + 4) * qscale, qh[rt][c], ql[rt][c]);
        }
    }

    const int kend = min(p.Sk, p.klen);
    const int kloop = CAUSAL ? min(kend, qlast + p.q_off + 1) : kend;       // keys the block's rows can see
    const int ntiles = (kloop + 63) >> 6;

    float m[RT][4], l[RT][4];
    f32x4 o[RT][NC];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { m[rt][r] = -INFINITY; l[rt][r] = 0.f; }
#pragma unroll
        for (int t = 0; t < NC; ++t) o[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- staging by role (wave-uniform): group 1 = K, 8 x 16-byte pieces per thread: piece i = row 8 i + ftid / 32, floats
    // 4 (ftid % 32) ..+4 (a wave instruction = two whole rows); group 0 = V, two 4-key x 4-column blocks per thread ----------------
    const unsigned ldk = (unsigned)p.ldk, ldv = (unsigned)p.ldv;
    const int klast = max(kend, 1) - 1;   // masked keys re-read the last VISIBLE row: rows in [klen, Sk) are caller memory (NaN bits there would turn p = 0 into NaN in the PV products)
    const int fk_key = ftid >> 5, fk_c = ftid & 31, fv_kg = ftid >> 4, fv_cg = ftid & 15;
    f32x4 st[8];
    auto row_of = [&](int key) __attribute__((always_inline)) {
        key = min(key, klast);
        if (PAGED) key = p.ktable[key >> 6] * 64 + (key & 63);
        return (unsigned)key;
    };
    // (Two parts each — K: keys 0-31 / 32-63 of the tile; V: columns 0-63 / 64-127.)  Ablations of the first form, conversion + LDS writes
    // in one block at the END of a tile with nothing to hide behind: 320 us at S = 2344, 209 without them (profiles/r04_attn_fa_ablate.txt).
    // Loads are unconditional (the tile index is clamped by the caller): hipcc counts vmcnt only for loads it can see on every path.
    auto load_part = [&](int part, int kt0) __attribute__((always_inline)) {
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                st[4 * part + i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Kb) +
                                                                  (__umul24(row_of(kt0 + 8 * (4 * part + i) + fk_key), ldk) + 4 * fk_c) * 4u);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                st[4 * part + u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Vb) +
                                                                  (__umul24(row_of(kt0 + 4 * fv_kg + u), ldv) + 4 * fv_cg + 64 * part) * 4u);
        }
    };
    auto store_part = [&](int part, unsigned char* buf) __attribute__((always_inline)) {
        if (kg == 1) {
            unsigned char* Kh = buf; unsigned char* Kl = buf + PL;
#pragma unroll
            for (int i = 4 * part; i < 4 * part + 4; ++i) {
                uint32_t h0, l0, h1, l1;
                at_split2(st[i][0], st[i][1], h0, l0);
                at_split2(st[i][2], st[i][3], h1, l1);
                const int key = 8 * i + fk_key;
                const int off = (fk_c >> 4) * 8192 + key * 128 + (((((fk_c & 15) >> 1)) ^ ((key >> 1) & 7)) << 4) + (fk_c & 1) * 8;
                *reinterpret_cast<uint2*>(Kh + off) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(Kl + off) = make_uint2(l0, l1);
            }
        } else {
            unsigned char* Vh = buf + 2 * PL; unsigned char* Vl = buf + 3 * PL;
            const int cg = fv_cg + 16 * part;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t h01, l01, h23, l23;
                at_split2(st[4 * part + 0][j], st[4 * part + 1][j], h01, l01);
                at_split2(st[4 * part + 2][j], st[4 * part + 3][j], h23, l23);
                const int d = 4 * cg + j;
                const int rho = 16 * (d % NC) + d / NC;
                const int off = rho * 128 + ((((fv_kg >> 1)) ^ ((rho >> 1) & 7)) << 4) + (fv_kg & 1) * 8;
                *reinterpret_cast<uint2*>(Vh + off) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(Vl + off) = make_uint2(l01, l23);
            }
        }
    };

    // IMG: tile `tile` of this KV head -> buffer `buf` (every thread moves 8 x 16 B; a wave instruction fills 1 KB of the image)
    const unsigned char* img_head = IMG ? p.kv_img + (size_t)hk * p.img_tiles * (4 * PL) : nullptr;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    auto dma_tile = [&](int tile, unsigned char* buf) __attribute__((always_inline)) {
        const unsigned char* src = img_head + (size_t)tile * (4 * PL);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa_glds16(src, (uint32_t)((i * 512 + tid) * 16), buf + (i * 512 + wu * 64) * 16);
    };
    if (IMG) {
        if (ntiles > 0) dma_tile(0, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (ntiles > 0) {
        load_part(0, 0); load_part(1, 0);
        store_part(0, lds); store_part(1, lds);
        const int k1 = ntiles > 1 ? 64 : 0;
        load_part(0, k1); load_part(1, k1);
    }
    __syncthreads();
    float* ps = patches + w * 16 * AT_PSTR3;
    for (int t = 0; t < ntiles; ++t) {
        const unsigned char* buf = lds + (t & 1) * 4 * PL;
        const unsigned char* Kh = buf; const unsigned char* Kl = buf + PL; const unsigned char* Vh = buf + 2 * PL; const unsigned char* Vl = buf + 3 * PL;
        const int kt0 = t * 64 + 32 * kg;
        unsigned char* nbuf = lds + ((t + 1) & 1) * 4 * PL;              // last read in tile t - 1: everyone is past it
        const bool more = t + 1 < ntiles;
        const int knext = min(t + 2, ntiles - 1) * 64;                  // (clamped: the last tile is simply re-read)
        if (IMG && more) dma_tile(t + 1, nbuf);                        // lands under this tile's products and softmax
        if (kt0 < kloop) {                                            // (wave-uniform) this half tile has a visible key
            // ---- S = Q K^T for the two 16-key sub-tiles ------------------------------------------------------------------------
            f32x4 sacc[RT][2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) sacc[rt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int key = 32 * kg + 16 * jt + lr;
#pragma unroll
                for (int c = 0; c < C32; ++c) {
                    const int gc = 4 * c + lg;
                    const int off = (gc >> 3) * 8192 + key * 128 + (((gc & 7) ^ ((key >> 1) & 7)) << 4);
                    const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Kh + off);      // one fragment read serves every row tile of the wave
                    const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Kl + off);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        f32x4 a = sacc[rt][jt];
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ql[rt][c], kh, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[rt][c], kl, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[rt][c], kh, a, 0, 0, 0);
                        sacc[rt][jt] = a;
                    }
                }
            }
            // conversion + LDS writes of the NEXT tile here, behind the 24 MFMAs of S that the matrix pipe is still working through, and the
            // reload for the tile after next right behind them: every load has a whole tile period to land.  (Splitting the staging in two
            // — half here, half behind the PV products — made hipcc wait vmcnt(0) here for the half reloaded a quarter tile earlier.)
            if (!IMG) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) { store_part(0, nbuf); store_part(1, nbuf); }
                load_part(0, knext); load_part(1, knext);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- online softmax in D layout (the direct kernel's) --------------------------------------------------------------
            const bool full_tile = kt0 + AT_KT <= (CAUSAL ? min(kend, q0 + p.q_off + 1) : kend);
            bf16x8 ph[RT], pl[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {                        // (the wave's ONE patch serves its row tiles in turn: LDS executes a wave's operations in order)
                float alpha[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + 16 * rt + lg * 4 + r;
                    float s0v = sacc[rt][0][r], s1v = sacc[rt][1][r];
                    if (!full_tile) {
                        const int klim = CAUSAL ? min(kend, q + p.q_off + 1) : kend;
                        s0v = (kt0 + lr < klim) ? s0v : -INFINITY;
                        s1v = (kt0 + 16 + lr < klim) ? s1v : -INFINITY;
                    }
                    const float mx = grp16_max(fmaxf(s0v, s1v));
                    const float mn = fmaxf(m[rt][r], mx);
                    const bool none = mn == -INFINITY;
                    alpha[r] = none ? 1.f : __builtin_amdgcn_exp2f(m[rt][r] - mn);
                    const float p0 = none ? 0.f : __builtin_amdgcn_exp2f(s0v - mn);
                    const float p1 = none ? 0.f : __builtin_amdgcn_exp2f(s1v - mn);
                    l[rt][r] = l[rt][r] * alpha[r] + grp16_sum(p0 + p1);
                    m[rt][r] = mn;
                    ps[(lg * 4 + r) * AT_PSTR3 + lr] = p0;
                    ps[(lg * 4 + r) * AT_PSTR3 + 16 + lr] = p1;
                }
#pragma unroll
                for (int tt = 0; tt < NC; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[rt][tt][r] *= alpha[r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                at_split8(*reinterpret_cast<const f32x4*>(ps + lr * AT_PSTR3 + 8 * lg), *reinterpret_cast<const f32x4*>(ps + lr * AT_PSTR3 + 8 * lg + 4), ph[rt], pl[rt]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // the patch is read before the next row tile (or the next key tile) rewrites it
            }
            // ---- O += P V over the 32 keys -------------------------------------------------------------------------------------
#pragma unroll
            for (int tt = 0; tt < NC; ++tt) {
                const int row = 16 * tt + lr;
                const int off = row * 128 + (((4 * kg + lg) ^ ((row >> 1) & 7)) << 4);
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Vh + off);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Vl + off);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    o[rt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[rt], bh, o[rt][tt], 0, 0, 0);
                    o[rt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[rt], bl, o[rt][tt], 0, 0, 0);
                    o[rt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[rt], bh, o[rt][tt], 0, 0, 0);
                }
            }
        } else if (!IMG) {
            if (more) { store_part(0, nbuf); store_part(1, nbuf); }
            load_part(0, knext); load_part(1, knext);
        }
        if (IMG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t + 1 have landed
        __syncthreads();
    }

    // ---- the second key half of every head merges into the first (every wave is past the last tile: the K / V buffers are free
    // and hold the hand-over) ---------------------------------------------------------------------------------------------------------
    {
        float* mg0 = reinterpret_cast<float*>(lds) + (size_t)(rg * 64 + lane) * RT * MGW;
        if (kg == 1) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float* mg = mg0 + rt * MGW;
#pragma unroll
                for (int r = 0; r < 4; ++r) { mg[r] = m[rt][r]; mg[4 + r] = l[rt][r]; }
#pragma unroll
                for (int t = 0; t < NC; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mg[8 + t * 4 + r] = o[rt][t][r];
            }
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float* mg = mg0 + rt * MGW;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m2 = mg[r], l2 = mg[4 + r];
                const float mn = fmaxf(m[rt][r], m2);
                const float a1 = (mn == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m[rt][r] - mn);
                const float a2 = (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
                l[rt][r] = l[rt][r] * a1 + l2 * a2;
                m[rt][r] = mn;
#pragma unroll
                for (int t = 0; t < NC; ++t) o[rt][t][r] = o[rt][t][r] * a1 + mg[8 + t * 4 + r] * a2;
            }
        }
    }

    float* Ob = p.O ? p.O + (size_t)b * p.bso + (size_t)h * D : nullptr;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + 16 * rt + lg * 4 + r;
        if (q >= p.Sq) continue;
        const float inv = (l[rt][r] > 0.f) ? 1.0f / l[rt][r] : 0.f;
#pragma unroll
        for (int j = 0; j < VQ; ++j) {
            const f32x4 v = f32x4{o[rt][4 * j][r] * inv, o[rt][4 * j + 1][r] * inv, o[rt][4 * j + 2][r] * inv, o[rt][4 * j + 3][r] * inv};
            if (Ob) *reinterpret_cast<f32x4*>(Ob + (size_t)q * p.ldo + lr * NC + 4 * j) = v;
            if (p.O_hi) {
                uint32_t hi[2], lo[2];
                at_split2(v[0], v[1], hi[0], lo[0]);
                at_split2(v[2], v[3], hi[1], lo[1]);
                const size_t at = ((size_t)b * p.Sq + q) * p.ldo_split + (size_t)h * D + lr * NC + 4 * j;
                *reinterpret_cast<uint2*>(p.O_hi + at) = make_uint2(hi[0], hi[1]);
                *reinterpret_cast<uint2*>(p.O_lo + at) = make_uint2(lo[0], lo[1]);
            }
        }
    }
}

}  // namespace

// would vhk_attn run the flash kernel for these arguments?  (the prefill asks before it spends a pass on the K / V tile images)
int vhk_attn_fa_applies(const VhAttnArgs& a) {
    if (a.Sq <= 0 || a.Sk <= 0 || a.P != nullptr || a.d != 128 || a.Hq != 4 * a.Hkv || a.chunk > 0) return 0;
    if (vh_tuning()->attn_impl != 0 || vh_tuning()->attn_fa == 0) return 0;
    const long blocks = (long)((a.Sq + 15) / 16) * a.Hkv * a.B;
    return (vh_tuning()->attn_fa == 2 || blocks * 2 >= vh_num_cus()) ? 1 : 0;
}

int vhk_attn(hipStream_t st, const VhAttnArgs& a_in) {
    VhAttnArgs a = a_in;
    a.xcd_map = vh_tuning()->attn_xcd != 0 ? 1 : 0;
    if (a.Sq <= 0 || a.Sk <= 0 || a.Hq % a.Hkv != 0) return -1;
    const bool rel = a.P != nullptr;
    const int want = vh_tuning()->attn_ksplit;
    const int kl = a.klen < a.Sk ? a.klen : a.Sk;
    // direct-operand kernel (default): every operand row must allow 16-byte loads / stores
    auto al16 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
    if (!a.O && !a.O_hi) return -1;
    if (a.O_hi && (!a.O_lo || (a.ldo_split % 4) != 0 || !al16(a.O_hi) || !al16(a.O_lo))) return -1;
    const bool direct_ok = al16(a.Q) && al16(a.K) && al16(a.V) && al16(a.O) && (!rel || al16(a.P)) &&
                           (!rel || (al16(a.bias_u) && al16(a.bias_v) && (a.ldp % 4) == 0 && (a.hsp % 4) == 0)) &&
                           ((a.ldq | a.hsq | a.bsq | a.ldk | a.hsk | a.bsk | a.ldv | a.hsv | a.ldo | a.bso) % 4) == 0;
    if (direct_ok && (a.d == 64 || (a.d == 128 && !rel))) {
        const dim3 g16((a.Sq + 15) / 16, a.Hq, a.B);
        int ks = want > 0 ? want : 4;
        if (kl <= AT_KT) ks = 1;
        else if (kl <= 2 * AT_KT && ks > 2) ks = 2;
        if (a.d == 128) ks = ks >= 2 ? 2 : 1;
        // waves per SIMD the register allocation aims at: 3 for the plain d = 64 kernel (<= 168 VGPRs: 768 resident 4-wave
        // blocks for the ViT's 1040 instead of 512), 2 elsewhere (rel-pos and d = 128 need > 200 registers)
#define AT_LAUNCH(DD, RR, KK) hipLaunchKernelGGL((k_attn_direct<DD, RR, KK, 2>), g16, dim3(64 * KK), 0, st, a)
#define AT_LAUNCH_W(DD, RR, KK, WW) hipLaunchKernelGGL((k_attn_direct<DD, RR, KK, WW>), g16, dim3(64 * KK), 0, st, a)
        // bf16 x 3 products (attn_impl 2: the fp32-MFMA kernel below).  Mask flavours are compile-time: pad mask only, causal,
        // causal + page table; anything else (chunk masks, a page table without causal) takes the fp32 kernel.
        int mode = a.chunk > 0 ? -1 : (!a.causal ? (a.ktable ? -1 : 0) : (a.ktable ? 2 : 1));
        {   // 32-bit byte offsets inside one head's K / V: rows (24-bit) x stride (24-bit) + a row, in bytes, below 2^32
            const long rows = a.ktable ? (a.kv_rows > 0 ? a.kv_rows : (1L << 24)) : a.Sk;
            const long ldmax = a.ldk > a.ldv ? a.ldk : a.ldv;
            if (rows >= (1L << 24) || ldmax >= (1L << 24) || ldmax < 0 || (rows * ldmax + a.d) * 4 >= (1L << 32)) mode = -1;
        }
        if (!rel && vh_tuning()->attn_impl == 0 && mode >= 0) {
            // flash form (K / V tiles shared through LDS by the four query heads of a KV head): d = 128 with the 4 : 1 grouping; taken when
            // its blocks (16 rows x 4 heads) still fill half the chip, or when forced (attn_fa = 2: tests)
            const int fa = vh_tuning()->attn_fa;
            if (fa != 0 && a.d == 128 && a.Hq == 4 * a.Hkv) {
                const dim3 gf((a.Sq + 15) / 16, a.Hkv, a.B);
                if (fa == 2 || (long)gf.x * gf.y * gf.z * 2 >= vh_num_cus()) {
                    // producer-side K / V tile images: one-shot causal prefills only (every key of the call was written by this pass)
                    const bool img = a.kv_img != nullptr && mode >= 1 && a.q_off == 0 && a.Sk == a.Sq && a.B == 1 && a.klen >= a.Sk &&
                                     a.img_tiles >= (a.Sk + 63) / 64 && (reinterpret_cast<uintptr_t>(a.kv_img) & 15) == 0;
                    // 32 query rows per wave (r06: every K / V fragment read from LDS feeds two row tiles, half the tile DMA and half the block
                    // barriers per query row; 2 waves per SIMD either way) when the launch still has two blocks per CU (S = 2344: 592 blocks;
                    // S = 552 would have 144 and keeps 16 rows); attn_rows = 16 / 32 forces one or the other
                    const dim3 gf2((a.Sq + 31) / 32, a.Hkv, a.B);
                    const int rows = vh_tuning()->attn_rows;
                    const bool two = mode == 1 && (rows == 32 || (rows == 0 && (long)gf2.x * gf2.y * gf2.z >= 2L * vh_num_cus()));
                    if (img && two) hipLaunchKernelGGL((k_attn_fa<1, true, 2>), gf2, dim3(512), 0, st, a);
                    else if (img) hipLaunchKernelGGL((k_attn_fa<1, true, 1>), gf, dim3(512), 0, st, a);
                    else if (mode == 0) hipLaunchKernelGGL((k_attn_fa<0, false, 1>), gf, dim3(512), 0, st, a);
                    else if (mode == 1 && two) hipLaunchKernelGGL((k_attn_fa<1, false, 2>), gf2, dim3(512), 0, st, a);
                    else if (mode == 1) hipLaunchKernelGGL((k_attn_fa<1, false, 1>), gf, dim3(512), 0, st, a);
                    else hipLaunchKernelGGL((k_attn_fa<2, false, 1>), gf, dim3(512), 0, st, a);
                    return 0;
                }
            }
            const dim3 g32((a.Sq + 31) / 32, a.Hq, a.B);
            // 32 rows per wave (K / V loaded and converted once for two row tiles): 57 against 63 us on the ViT once the loop
            // was VALU-lean; only when the launch still gives every SIMD two waves
            const int rows = vh_tuning()->attn_rows;
            const bool two = a.d == 64 && mode == 0 && (rows == 32 || (rows == 0 && (long)g32.x * g32.y * g32.z * ks >= 8L * vh_num_cus()));
#define X3(DD, KK, WW, RR, MM, G) hipLaunchKernelGGL((k_attn_x3<DD, KK, WW, RR, MM>), G, dim3(64 * KK), 0, st, a)
#define X3_MODES(DD, KK, WW) do { if (mode == 0) X3(DD, KK, WW, 1, 0, g16); else if (mode == 1) X3(DD, KK, 2, 1, 1, g16); else X3(DD, KK, 2, 1, 2, g16); } while (0)
            if (two) {
                if (ks == 1) X3(64, 1, 2, 2, 0, g32); else if (ks == 2) X3(64, 2, 2, 2, 0, g32); else X3(64, 4, 2, 2, 0, g32);
            } else if (a.d == 64) {
                if (ks == 1) X3_MODES(64, 1, 2); else if (ks == 2) X3_MODES(64, 2, 2); else X3_MODES(64, 4, 2);
            } else {
                if (ks == 1) X3_MODES(128, 1, 2); else X3_MODES(128, 2, 2);
            }
#undef X3_MODES
#undef X3
            return 0;
        }
        if (a.d == 64 && !rel) {
            if (ks == 1) AT_LAUNCH(64, false, 1);
            else if (ks == 2) AT_LAUNCH(64, false, 2);
            else AT_LAUNCH_W(64, false, 4, 3);
        }
        else if (a.d == 64) { if (ks == 1) AT_LAUNCH(64, true, 1); else if (ks == 2) AT_LAUNCH(64, true, 2); else AT_LAUNCH(64, true, 4); }
        else { if (ks == 1) AT_LAUNCH(128, false, 1); else AT_LAUNCH(128, false, 2); }
#undef AT_LAUNCH
#undef AT_LAUNCH_W
        return 0;
    }
    return -1;   // rows that do not allow 16-byte accesses, d outside {64, 128}, rel-pos at d = 128: rejected (VH_E_SHAPE)
}
