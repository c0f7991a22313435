// vh_gemm.hip — the multi-row GEMM workhorse (SURVEY §2.4 K1,K4,K6,K7,K9,K12,K13,K16,K17,
// K21,K24,K26-prefill): C = epilogue(A · W^T) with fp32 activations A, bf16 weights W.
//
// MFMA: v_mfma_f32_16x16x32_bf16.  The fp32 activation tile is split once, while it is
// staged into LDS, into hi = bf16(a) and lo = bf16(a - hi); each weight fragment is then
// used by two MFMAs (hi and lo) accumulating into the same fp32 tile, i.e. the GEMM sees
// 16 mantissa bits of A at the cost of 2x matrix-core work and no extra weight traffic.
// (Prefill at S<~1000 and the encoders are weight-read/latency bound before they are
// MFMA bound, so this buys fp32-oracle parity for little wall time.)
//
// Tile: 256 threads = 4 waves (2 along M x 2 along N); block tile 64(M) x 128(N) x 64(K);
// wave tile 32 x 64 = 2x4 MFMA tiles (8 accumulators).  LDS rows are 128 B (64 bf16) with
// the 16-byte chunk index XOR-swizzled by (row>>1)&7 so that every ds_read_b128 lane group
// (rows {0-3,12-15} at chunk c and rows {4-11} at chunk c^1) hits 16 distinct 16-B slots.
// Global->LDS is register staged and issued one K-tile ahead of the MFMAs (two for the plain GEMMs).
//
// A addressing is generalised so convolutions and the MoE gather need no im2col copy:
//   source row(m, k) = a_rowidx[m] (or m) + segrow[k / seglen],  column = k % seglen.
// Grouped mode (top-2 MoE): rows are pre-sorted by expert, group_off[e..e+1] bounds expert
// e's rows and W advances by w_group_stride per expert; the 1-D grid enumerates (expert,
// m-tile) pairs on device so no host sync is needed to size the launch.
#include "vh_common.h"
#include "vh_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

#define GM_BM 64
#define GM_BN 128
#define GM_BK 64

__device__ __forceinline__ int lds_off(int row, int chunk) {  // byte offset in a [rows][128 B] tile
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// 256 threads, waves 2 (M) x 2 (N), block tile 64 x 128 (GLU: 64 gate + 64 up weight rows -> 64 outputs).
// Tile shapes that trade occupancy for size were all measured slower on this structure (DESIGN.md §6.2:
// 128- and 256-row m-tiles, pre-split operand planes, XCD-contiguous block orders) and were removed again.
// (r04 re-measured 128 x 128 tiles on the encoders' plain Linears, where they turn the ViT's 408-block qkv launch into 216 blocks in
// one round: 58 vs 41.5 us, ViT + projector 6.9 vs 5.3 ms — at one 4-wave block per CU nothing hides the K-tile round trip;
// profiles/r04_gemm_tall_ab.txt.)
// MINW pins the register-allocation target: hipcc otherwise chases the occupancy the 32 KB of LDS would
// allow (5 blocks/CU) and parks prefetched weight registers in SCRATCH to get under ~96 VGPRs, which turns
// the asynchronous prefetch into a synchronous round trip.
template <bool GLU, int PF, int MINW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MINW, MINW)))
void k_gemm(const VhGemmArgs p) {
    constexpr int BM = GM_BM;
    constexpr int WMW = 2;                           // waves along M (2 along N)
    constexpr int MI = BM / WMW / 16;                // 16-row tiles per wave
    constexpr int AF4 = BM * 16 / 256;               // 16-byte pieces of A per thread per K-tile (4)
    constexpr int ATPR = 16 / AF4;                   // threads per A row (4)
    constexpr int WU4 = GM_BN * 8 / 256;             // 16-byte pieces of W per thread per K-tile (4)
    constexpr int WTPR = 8 / WU4;                    // threads per W row (2)
    __shared__ __attribute__((aligned(16))) unsigned char lds_ahi[BM * 128];
    __shared__ __attribute__((aligned(16))) unsigned char lds_alo[BM * 128];
    __shared__ __attribute__((aligned(16))) unsigned char lds_w[GM_BN * 128];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid % WMW, wn = wid / WMW;
    constexpr int NT = GLU ? 64 : 128;  // output columns per block

    // ---- which (n-tile, group, m-tile) is this block?  1-D grid, n-tiles fastest: blocks that run
    // together share the activation tile, which stays L2-resident ------------------------------------
    const int mt = p.mt_slots;
    const int ntl = gridDim.x / mt;
    const int slot = blockIdx.x / ntl, n_tile = blockIdx.x - slot * ntl;
    int m_begin, m_end;
    const uint16_t* Wb = p.W;
    const uint16_t* Wu = p.W_up;
    if (p.group_off) {
        int tile = slot, e = 0;
        for (; e < p.ngroups; ++e) {
            const int cnt = p.group_off[e + 1] - p.group_off[e];
            const int nt = (cnt + BM - 1) / BM;
            if (tile < nt) break;
            tile -= nt;
        }
        if (e == p.ngroups) return;
        m_begin = p.group_off[e] + tile * BM;
        m_end = p.group_off[e + 1];
        Wb += (size_t)e * p.w_group_stride;
        if (GLU) Wu += (size_t)e * p.w_group_stride;
    } else {
        m_begin = slot * BM;
        m_end = p.M;
        if (m_begin >= m_end) return;
    }
    const int n_begin = n_tile * NT;
    const int rows_here = min(m_end - m_begin, BM);
    const int nrt = (rows_here + 15) >> 4;           // 16-row tiles that hold data

    // ---- staging assignments -------------------------------------------------------
    const int arow = tid / ATPR, achunk0 = (tid % ATPR) * (AF4 / 2);  // AF4/2 chunks of 8 floats
    const int am = m_begin + arow;
    const bool a_valid_m = am < m_end;
    const int a_src = a_valid_m ? (p.a_rowidx ? p.a_rowidx[am] : am) : 0;

    const int wrow = tid / WTPR, wchunk0 = (tid % WTPR) * WU4;
    const uint16_t* wptr;
    bool w_valid;
    if (GLU) {
        const int n = n_begin + (wrow & 63);
        w_valid = n < p.N;
        wptr = ((wrow < 64) ? Wb : Wu) + (size_t)(w_valid ? n : 0) * p.ldw + wchunk0 * 8;
    } else {
        const int n = n_begin + wrow;
        w_valid = n < p.N;
        wptr = Wb + (size_t)(w_valid ? n : 0) * p.ldw + wchunk0 * 8;
    }

    uint4 ra0[AF4], ra1[AF4];     // raw 16-byte pieces of the A row (fp32 x4)
    u32x4 rw0[WU4], rw1[WU4];    // native vector type: the HIP uint4 struct kept this array in scratch memory
    uint32_t keep0 = 0, keep1 = 0;   // all-ones where the logical A row exists (applied when the tile is consumed)
    auto load_tile = [&](int kt, uint4 (&ra)[AF4], u32x4 (&rw)[WU4], uint32_t& keep) __attribute__((always_inline)) {
        const int k0 = kt * GM_BK;
        const int seg = k0 / p.seglen;
        const int koff = k0 - seg * p.seglen;
        const int srow = a_src + p.segrow[seg];
        // UNCONDITIONAL loads from clamped addresses, masked to zero WHEN CONSUMED where the logical row
        // does not exist: with the loads under a branch hipcc cannot count them and waits vmcnt(0) before every
        // use, which drains the NEXT tile's prefetch too (seen in the ISA of the two-tile variant).
        const bool a_ok = a_valid_m && srow >= 0 && srow < p.a_rows;
        const size_t a_at = (size_t)(a_ok ? srow : 0) * p.lda + koff + achunk0 * 8;
        keep = a_ok ? 0xffffffffu : 0u;
        const uint4* ap = reinterpret_cast<const uint4*>(p.A + a_at);
#pragma unroll
        for (int i = 0; i < AF4; ++i) ra[i] = ap[i];
        const u32x4* wp = reinterpret_cast<const u32x4*>(wptr + k0);   // rows past N are clamped; their columns are never stored
#pragma unroll
        for (int i = 0; i < WU4; ++i) rw[i] = wp[i];
    };
    auto store_tile = [&](const uint4 (&ra)[AF4], const u32x4 (&rw)[WU4], const uint32_t keep) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < AF4 / 2; ++c) {
            const uint4 f0 = ra[c * 2], f1 = ra[c * 2 + 1];
            const uint32_t v[8] = {f0.x & keep, f0.y & keep, f0.z & keep, f0.w & keep,
                                   f1.x & keep, f1.y & keep, f1.z & keep, f1.w & keep};
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) split_bf16_pair(__uint_as_float(v[2 * i]), __uint_as_float(v[2 * i + 1]), hi[i], lo[i]);
            const uint4 ph = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            const uint4 pl = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            const int off = lds_off(arow, achunk0 + c);
            *reinterpret_cast<uint4*>(lds_ahi + off) = ph;
            *reinterpret_cast<uint4*>(lds_alo + off) = pl;
        }
#pragma unroll
        for (int c = 0; c < WU4; ++c) *reinterpret_cast<u32x4*>(lds_w + lds_off(wrow, wchunk0 + c)) = rw[c];
    };

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mfma_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 4 + (lane >> 4);
            bf16x8_t bw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int r;
                if (GLU) r = (j < 2) ? (wn * 32 + j * 16) : (64 + wn * 32 + (j - 2) * 16);
                else r = wn * 64 + j * 16;
                bw[j] = *reinterpret_cast<const bf16x8_t*>(lds_w + lds_off(r + (lane & 15), chunk));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int rt = wm + WMW * i;          // row tiles interleaved over the M waves
                if (rt < nrt) {                       // wave-uniform
                    const int off = lds_off(rt * 16 + (lane & 15), chunk);
                    const bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(lds_ahi + off);
                    const bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(lds_alo + off);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bw[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bw[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };

    // Two K-tiles of operands are kept in flight in registers (sets 0 / 1): a K-tile iteration is
    // otherwise bounded by one global-load round trip (~1-2 us under load), which is what made every
    // small-M GEMM of the encoders cost ~2 us per 64 of K regardless of its size.
    // split-K: blockIdx.y takes the K-tiles [kt_lo, nkt) of its share and writes raw partial sums to its slab
    const int nkt_all = p.K / GM_BK;
    const int KSP = gridDim.y;
    const int kt_lo = (int)(((long)nkt_all * blockIdx.y) / KSP);
    const int nkt = (int)(((long)nkt_all * (blockIdx.y + 1)) / KSP);
    // loads are issued UNCONDITIONALLY (the tile index is clamped, the last tile is simply re-read): a
    // load under `if (kt + PF < nkt)` makes the number of outstanding loads unknown to hipcc, which then
    // waits vmcnt(0) and drains the younger tile as well
    load_tile(kt_lo, ra0, rw0, keep0);
    if (PF == 2) load_tile(min(kt_lo + 1, nkt - 1), ra1, rw1, keep1);
    for (int kt = kt_lo; kt < nkt; kt += PF) {
        __syncthreads();  // previous tile's fragment reads are done
        store_tile(ra0, rw0, keep0);
        __syncthreads();
        load_tile(min(kt + PF, nkt - 1), ra0, rw0, keep0);  // in flight during the MFMAs below (and the next tile)
        mfma_tile();
        if (PF == 2) {
            if (kt + 1 >= nkt) break;
            __syncthreads();
            store_tile(ra1, rw1, keep1);
            __syncthreads();
            load_tile(min(kt + 3, nkt - 1), ra1, rw1, keep1);
            mfma_tile();
        }
    }

    // ---- epilogue: D layout col = lane&15, row = (lane>>4)*4 + r ---------------------
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int rt = wm + WMW * i;
        if (rt >= nrt) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m_begin + rt * 16 + (lane >> 4) * 4 + r;
            if (m >= m_end) continue;
            const long orow = p.c_rowidx ? p.c_rowidx[m] : m;
            if (GLU) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n_begin + wn * 32 + j * 16 + (lane & 15);
                    if (n < p.N) p.C[orow * p.ldc + n] = silu_f(acc[i][j][r]) * acc[i][j + 2][r];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n_begin + wn * 64 + j * 16 + (lane & 15);
                    if (n < p.N && KSP > 1) {
                        p.ws[((size_t)blockIdx.y * p.M + m) * p.N + n] = acc[i][j][r];     // epilogue in k_gemm_reduce
                    } else if (n < p.N) {
                        float v = acc[i][j][r];
                        if (p.bias) v += p.bias[n];
                        v = apply_act(v, p.act);
                        if (p.scale) v *= p.scale[n];
                        if (p.resid) v += p.resid[orow * p.ldr + n];
                        p.C[orow * p.ldc + n] = v;
                    }
                }
            }
        }
    }
}

// sum of the split-K slabs + the GEMM epilogue (4 columns per thread)
__global__ __launch_bounds__(256) void k_gemm_reduce(const VhGemmArgs p, int ksp) {
    const int n4 = p.N >> 2;
    const long total = (long)p.M * n4;
    const size_t slab = (size_t)p.M * p.N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / n4;
        const int n = (int)(i - m * n4) * 4;
        const float* src = p.ws + (size_t)m * p.N + n;
        f32x4 v = *reinterpret_cast<const f32x4*>(src);
        for (int k = 1; k < ksp; ++k) v += *reinterpret_cast<const f32x4*>(src + k * slab);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = v[j];
            if (p.bias) x += p.bias[n + j];
            x = apply_act(x, p.act);
            if (p.scale) x *= p.scale[n + j];
            if (p.resid) x += p.resid[m * p.ldr + n + j];
            p.C[m * p.ldc + n + j] = x;
        }
    }
}

// The same reduction with one BLOCK per output row (N <= 4096), followed by the LayerNorm of that row (VhGemmArgs::ln_*):
// the encoders' Linear -> (+ bias, layer scale, residual) -> LayerNorm chains (InternViT: modeling_intern_vit.py:245-253,
// Whale: transformer.py encoder layer) cost one launch instead of reducer + norm.  Thread t owns the 16-byte chunks
// t + 256 j; all slab loads of a chunk are issued before they are added (a first version with one wave per row and a
// serial slab loop took 14-27 us against 5 + 7 for the two kernels it replaces: profiles/r03_encoder_pass_trace.txt).
#define GR_MAXJ 4
template <int KSP>
__global__ __launch_bounds__(256) void k_gemm_reduce_ln(const VhGemmArgs p, int ksp_rt) {
    __shared__ float red[4];
    const long m = blockIdx.x;
    const int nv = p.N >> 2;
    const int ksp = KSP > 0 ? KSP : ksp_rt;
    const size_t slab = (size_t)p.M * p.N;
    f32x4 v[GR_MAXJ];
    float s[1] = {0.f};
#pragma unroll
    for (int j = 0; j < GR_MAXJ; ++j) {
        const int c = threadIdx.x + j * 256;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c >= nv) continue;
        const int n = c * 4;
        const float* src = p.ws + (size_t)m * p.N + n;
        f32x4 a = *reinterpret_cast<const f32x4*>(src);
        if (KSP > 0) {
            f32x4 part[KSP > 1 ? KSP - 1 : 1];
#pragma unroll
            for (int k = 1; k < KSP; ++k) part[k - 1] = *reinterpret_cast<const f32x4*>(src + k * slab);
#pragma unroll
            for (int k = 1; k < KSP; ++k) a += part[k - 1];
        } else {
            for (int k = 1; k < ksp; ++k) a += *reinterpret_cast<const f32x4*>(src + k * slab);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x = a[i];
            if (p.bias) x += p.bias[n + i];
            x = apply_act(x, p.act);
            if (p.scale) x *= p.scale[n + i];
            if (p.resid) x += p.resid[m * p.ldr + n + i];
            a[i] = x;
        }
        *reinterpret_cast<f32x4*>(p.C + m * p.ldc + n) = a;
        v[j] = a;
        s[0] += (a[0] + a[1]) + (a[2] + a[3]);
    }
    block256_sum<1>(s, red);
    const float mean = s[0] / (float)p.N;
    float q[1] = {0.f};
#pragma unroll
    for (int j = 0; j < GR_MAXJ; ++j) {
        if (threadIdx.x + j * 256 < nv) {
            const f32x4 d = v[j] - mean;
            q[0] += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    block256_sum<1>(q, red);
    const float inv = rsqrtf(q[0] / (float)p.N + p.ln_eps);
#pragma unroll
    for (int j = 0; j < GR_MAXJ; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c >= nv) continue;
        const f32x4 ww = reinterpret_cast<const f32x4*>(p.ln_w)[c];
        f32x4 r = (v[j] - mean) * inv * ww;
        if (p.ln_b) r += reinterpret_cast<const f32x4*>(p.ln_b)[c];
        if (p.ln_out) *reinterpret_cast<f32x4*>(p.ln_out + m * p.ld_ln + c * 4) = r;
        if (p.ln_hi) {
            uint32_t h0, l0, h1, l1;
            split_bf16_pair(r[0], r[1], h0, l0);
            split_bf16_pair(r[2], r[3], h1, l1);
            *reinterpret_cast<uint2*>(p.ln_hi + m * p.ld_ln_split + c * 4) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(p.ln_lo + m * p.ld_ln_split + c * 4) = make_uint2(l0, l1);
        }
    }
}

}  // namespace

static void launch_reduce_ln(hipStream_t st, const VhGemmArgs& g, int ksp) {
    switch (ksp) {
        case 2: hipLaunchKernelGGL(k_gemm_reduce_ln<2>, dim3(g.M), dim3(256), 0, st, g, ksp); break;
        case 3: hipLaunchKernelGGL(k_gemm_reduce_ln<3>, dim3(g.M), dim3(256), 0, st, g, ksp); break;
        case 4: hipLaunchKernelGGL(k_gemm_reduce_ln<4>, dim3(g.M), dim3(256), 0, st, g, ksp); break;
        case 6: hipLaunchKernelGGL(k_gemm_reduce_ln<6>, dim3(g.M), dim3(256), 0, st, g, ksp); break;
        case 8: hipLaunchKernelGGL(k_gemm_reduce_ln<8>, dim3(g.M), dim3(256), 0, st, g, ksp); break;
        default: hipLaunchKernelGGL(k_gemm_reduce_ln<0>, dim3(g.M), dim3(256), 0, st, g, ksp); break;
    }
}

int vhk_gemm_reduce(hipStream_t st, const VhGemmArgs& a, int ksp) {
    if (!a.ws || !a.C || ksp < 1 || ksp > 8 || a.M < 0 || a.N <= 0 || (a.N % 4) != 0 || (a.ldc % 4) != 0) return -1;
    if ((size_t)ksp * a.M * a.N * sizeof(float) > a.ws_bytes) return -1;
    if (a.M == 0) return 0;
    const bool ln = a.ln_w != nullptr;
    if (ln && ((!a.ln_out && !a.ln_hi) || (a.ln_hi && (!a.ln_lo || (a.ld_ln_split % 4) != 0)) || (a.ln_out && (a.ld_ln % 4) != 0))) return -1;
    if (ln && a.N <= GR_MAXJ * 1024) { launch_reduce_ln(st, a, ksp); return 0; }
    if (ln && a.ln_hi) return -1;                      // planes come from the fused reducer only
    long rg = ((long)a.M * (a.N / 4) + 255) / 256;
    if (rg > 2048) rg = 2048;
    hipLaunchKernelGGL(k_gemm_reduce, dim3((int)rg), dim3(256), 0, st, a, ksp);
    if (ln) return vhk_layernorm(st, a.C, a.ldc, a.ln_out, a.ld_ln, a.ln_w, a.ln_b, a.M, a.N, a.ln_eps, VH_ACT_NONE, 1.0f);
    return 0;
}

int vhk_gemm(hipStream_t st, const VhGemmArgs& a) {
    if (a.K <= 0 || a.K % GM_BK != 0 || a.nseg < 1 || a.nseg > 16 || a.seglen % GM_BK != 0 ||
        a.nseg * a.seglen != a.K || a.M < 0 || a.N <= 0 || a.A == nullptr)
        return -1;
    if (a.ln_hi) return -1;                        // plane outputs: vhk_gemm_reduce only
    if (a.ln_out && (!a.ln_w || a.W_up || a.c_rowidx || a.group_off || (a.N % 4) != 0 || (a.ldc % 4) != 0 || (a.ld_ln % 4) != 0)) return -1;
    if (a.M == 0) return 0;
    VhGemmArgs g = a;
    g.mt_slots = a.group_off ? (a.M / GM_BM + a.ngroups) : (a.M + GM_BM - 1) / GM_BM;  // grouped: upper bound
    // split-K when the launch would leave most CUs with at most one 4-wave block (nothing to overlap a K-tile's load ->
    // split -> LDS -> barrier chain with): aim at ~3 blocks per CU, keep >= 4 K-tiles per block
    int ksp = 1;
    const bool plain = !a.group_off && !a.W_up && !a.c_rowidx && (a.N % 4) == 0 && a.ws != nullptr;
    if (plain && a.ksplit != 1) {
        const long blocks = (long)((a.N + GM_BN - 1) / GM_BN) * g.mt_slots;
        const int nkt = a.K / GM_BK;
        const long target = 3L * vh_num_cus();
        // (measured on the ViT: 136-block launches gain 1.5-1.8x, the 408-block qkv GEMM loses 20 % to the reducer)
        ksp = a.ksplit > 1 ? a.ksplit : (blocks * 4 >= 5L * vh_num_cus() ? 1 : (int)((target + blocks - 1) / blocks));
        if (ksp > nkt / 4) ksp = nkt / 4;
        if (ksp > 8) ksp = 8;
        while (ksp > 1 && (size_t)ksp * a.M * a.N * sizeof(float) > a.ws_bytes) --ksp;
        if (ksp < 1) ksp = 1;
    }
    const dim3 grid_glu(((a.N + 63) / 64) * g.mt_slots), grid(((a.N + GM_BN - 1) / GM_BN) * g.mt_slots, ksp);
    // two K-tiles in flight pays for the small plain GEMMs (encoders: -5..-9 %), not for the grouped ones (+-0)
    const bool pf2 = a.group_off == nullptr;
    if (a.W_up) {
        if (pf2) hipLaunchKernelGGL((k_gemm<true, 2, 3>), grid_glu, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((k_gemm<true, 1, 4>), grid_glu, dim3(256), 0, st, g);
    } else {
        if (pf2) hipLaunchKernelGGL((k_gemm<false, 2, 3>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((k_gemm<false, 1, 4>), grid, dim3(256), 0, st, g);
        const bool ln_fused = a.ln_out && ksp > 1 && a.N <= GR_MAXJ * 1024 && (a.ldc % 4) == 0 && (a.ld_ln % 4) == 0;
        if (ksp > 1 && ln_fused) {
            launch_reduce_ln(st, g, ksp);
            return 0;
        }
        if (ksp > 1) {
            long rg = ((long)a.M * (a.N / 4) + 255) / 256;
            if (rg > 2048) rg = 2048;
            hipLaunchKernelGGL(k_gemm_reduce, dim3((int)rg), dim3(256), 0, st, g, ksp);
        }
    }
    if (a.ln_out) return vhk_layernorm(st, a.C, a.ldc, a.ln_out, a.ld_ln, a.ln_w, a.ln_b, a.M, a.N, a.ln_eps, VH_ACT_NONE, 1.0f);
    return 0;
}
