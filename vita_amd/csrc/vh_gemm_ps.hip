// vh_gemm_ps.hip — skinny-M GEMM on PRE-SPLIT activations for the Mixtral prefill (SURVEY §2.4 K26:
// top-2 MoE grouped GEMMs; HF MixtralExperts, modeling_mixtral.py:57-93).
//
// Why a second GEMM kernel.  At prefill lengths of a few hundred tokens every expert sees ~S/4 rows
// (138 at S=552): the job is a weight stream with a short M, and rocprof showed the general kernel
// (vh_gemm.hip, 64x128 tiles, register-staged) latency-bound on its operand loads (27 % MFMA busy,
// W re-fetched once per 64-row m-tile).  This kernel is shaped for that regime:
//   * ONE m-tile of up to 192 rows covers all rows of an expert, so each weight byte leaves HBM once;
//     only ceil(rows/16) row tiles are multiplied (dynamic, wave-uniform), 138 rows cost 9/12 of a tile;
//   * activations arrive already split into bf16 hi/lo planes (x = hi + lo to 2^-17, the library's
//     exact-mode contract, vh_common.h) — the split is done once by the producer instead of once per
//     n-tile, and operand staging becomes a pure copy;
//   * all three operand tiles (A_hi, A_lo, W) go global -> LDS with global_load_lds (16 B/lane, no
//     VGPR round trip) in whole 128-B lines, two BK=64 stages (2 x 80 KiB = the whole LDS): stage k+1 is in
//     flight while stage k is multiplied; raw s_barrier, one per stage;
//   * LDS image: 16-row x 128-B sub-tiles, each written lane-linearly by two glds (8 rows each); bank
//     conflicts are removed by permuting the 16-B chunks of a row on the SOURCE side (chunk ^ (row>>1)),
//     which makes every ds_read_b128 lane group hit 16 distinct 16-B slots;
//   * waves: 2 (M) x WN (N); a wave owns row tiles wm, wm+2, ... and 64 weight rows (GLU: 32 gate +
//     32 up rows, SiLU(g)*u in the epilogue); MFMA v_mfma_f32_16x16x32_bf16, hi and lo both
//     accumulate into the same fp32 tile.
// 1-D grid, remapped so each XCD owns a contiguous run of tiles that share the same activation rows.
#include "vh_common.h"
#include "vh_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((address_space(3))) void* lds_void_t;
typedef const __attribute__((address_space(1))) void* glb_void_t;

#define PS_BM 192
#define PS_RT 12                     // 16-row tiles per block
#define PS_SUB 2048                  // one 16-row x 128-byte (BK = 64) sub-tile
#define PS_PLANE (PS_RT * PS_SUB)    // bytes of one activation plane per stage

__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_wave_base) {
    // lane l's 16 bytes land at lds_wave_base + 16*l (wave-uniform base goes through M0)
    __builtin_amdgcn_global_load_lds((glb_void_t)gsrc, (lds_void_t)lds_wave_base, 16, 0, 0);
}

template <int WN, bool GLU>
__global__ __launch_bounds__(128 * WN) void k_gemm_ps(const VhGemmPsArgs p) {
    constexpr int NWAVES = 2 * WN;
    constexpr int BNW = 64 * WN;                    // weight rows per block
    constexpr int NOUT = GLU ? BNW / 2 : BNW;       // output columns per block
    constexpr int W_BYTES = BNW * 128;
    constexpr int STAGE = 2 * PS_PLANE + W_BYTES;   // 80 KiB (WN=4) / 64 KiB (WN=2)
    constexpr int GA = (2 * PS_RT * 2) / NWAVES;    // activation glds (8 rows x 128 B each) per wave per stage
    constexpr int GW = (BNW / 8) / NWAVES;          // weight glds per wave per stage
    static_assert((4 * PS_RT) % NWAVES == 0 && (BNW / 8) % NWAVES == 0, "glds split");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid & 1, wn = wid >> 1;

    // ---- tile of this block ---------------------------------------------------------------------
    const int n_tiles = (p.N + NOUT - 1) / NOUT;
    int l;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, idx = b >> 3, q = nwg >> 3, r = nwg & 7;
        l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int slot = l / n_tiles, n_tile = l - slot * n_tiles;
    int m_begin, m_end;
    const uint16_t* Wb = p.W;
    const uint16_t* Wu = p.W_up;
    if (p.group_off) {
        int tile = slot, e = 0;
        for (; e < p.ngroups; ++e) {
            const int cnt = p.group_off[e + 1] - p.group_off[e];
            const int nt = (cnt + PS_BM - 1) / PS_BM;
            if (tile < nt) break;
            tile -= nt;
        }
        if (e == p.ngroups) return;
        m_begin = p.group_off[e] + tile * PS_BM;
        m_end = p.group_off[e + 1];
        Wb += (size_t)e * p.w_group_stride;
        if (GLU) Wu += (size_t)e * p.w_group_stride;
    } else {
        m_begin = slot * PS_BM;
        m_end = p.M;
        if (m_begin >= m_end) return;
    }
    if (m_end - m_begin > PS_BM) m_end = m_begin + PS_BM;
    const int nrt = (m_end - m_begin + 15) >> 4;    // row tiles that hold data

    // ---- per-lane source pointers of this wave's glds.  One glds moves 8 rows x 128 B (whole cache
    // lines: 64-B pieces ran the HBM/L2 side at 2.4 TB/s).  Lane l reads row l>>3; its 16-B chunk is
    // permuted inside the row, chunk = (l&7) ^ (row16>>1), which is what makes the fragment reads
    // below bank-conflict free.  Fixed glds count per stage: rows past the tile are clamped to a valid
    // row (their products are never stored). ------------------------------------------------------
    const int lrow = lane >> 3;
    const uint16_t* a_src[GA];
    int a_off[GA];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int q = wid + NWAVES * i;                    // 0 .. 4*PS_RT-1
        const int plane = q / (2 * PS_RT), qs = q % (2 * PS_RT);
        const int s = qs >> 1, half = qs & 1;              // sub-tile, 8-row half
        const int r16 = half * 8 + lrow;
        int m = m_begin + s * 16 + r16;
        if (m > m_end - 1) m = m_end - 1;
        const long src_row = p.a_rowidx ? p.a_rowidx[m] : m;
        const int c8 = (lane & 7) ^ ((r16 >> 1) & 7);
        a_src[i] = (plane ? p.A_lo : p.A_hi) + (size_t)src_row * p.lda + c8 * 8;
        a_off[i] = plane * PS_PLANE + s * PS_SUB + half * 1024;
    }
    const uint16_t* w_src[GW];
    int w_off[GW];
#pragma unroll
    for (int i = 0; i < GW; ++i) {
        const int g = wid + NWAVES * i;                    // 8-row group of the block's weight rows
        const int R = g * 8 + lrow;
        int n;
        const uint16_t* base = Wb;
        if (GLU) {
            n = n_tile * NOUT + (R >> 6) * 32 + (R & 31);
            if (R & 32) base = Wu;
        } else {
            n = n_tile * NOUT + R;
        }
        if (n > p.N - 1) n = p.N - 1;
        const int c8 = (lane & 7) ^ (((R & 15) >> 1) & 7);
        w_src[i] = base + (size_t)n * p.ldw + c8 * 8;
        w_off[i] = 2 * PS_PLANE + g * 1024;
    }
    auto issue = [&](int kt, int buf) {
        unsigned char* sb = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < GA; ++i)
            if (!(p.ablate & 1)) glds16(a_src[i] + (size_t)kt * 64, sb + a_off[i]);
#pragma unroll
        for (int i = 0; i < GW; ++i)
            if (!(p.ablate & 2)) glds16(w_src[i] + (size_t)kt * 64, sb + w_off[i]);
    };

    // fragment read offset inside a sub-tile for k-step ks: row r = lane&15, chunk = ks*4 + (lane>>4)
    const int fr = lane & 15;
    const int frag_base = (fr >> 3) * 1024 + (fr & 7) * 128;
    const int frag_x = (fr >> 1) & 7;
    auto frag_off = [&](int ks) { return frag_base + (((ks * 4 + (lane >> 4)) ^ frag_x) << 4); };

    f32x4 acc[PS_RT / 2][4];
#pragma unroll
    for (int t = 0; t < PS_RT / 2; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K >> 6;
    issue(0, 0);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of stage kt has landed
        __builtin_amdgcn_s_barrier();                      // ... and everybody's; stage kt-1 is fully consumed
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);           // in flight during the MFMAs below
        const unsigned char* sb = lds + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fo = frag_off(ks);
            // all fragment reads of the k-step first (unconditional: rows past the tile hold clamped
            // copies), then the MFMA groups back to back
            bf16x8_t bw[4], ah[PS_RT / 2], al[PS_RT / 2];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                bw[c] = *reinterpret_cast<const bf16x8_t*>(sb + 2 * PS_PLANE + (wn * 4 + c) * PS_SUB + fo);
#pragma unroll
            for (int t = 0; t < PS_RT / 2; ++t) {
                const int rt = wm + 2 * t;
                ah[t] = *reinterpret_cast<const bf16x8_t*>(sb + rt * PS_SUB + fo);
                al[t] = *reinterpret_cast<const bf16x8_t*>(sb + PS_PLANE + rt * PS_SUB + fo);
            }
#pragma unroll
            for (int t = 0; t < PS_RT / 2; ++t) {
                if (wm + 2 * t < nrt && !(p.ablate & 4)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bw[c], acc[t][c], 0, 0, 0);
                        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t], bw[c], acc[t][c], 0, 0, 0);
                    }
                }
            }
        }
        buf ^= 1;
    }

    // ---- epilogue -------------------------------------------------------------------------------
    // MFMA D layout (col = lane&15, row = (lane>>4)*4 + r) -> LDS fp32 tile -> row-contiguous wide
    // stores (16 B per lane; 2-byte D-layout stores were measured at ~55 % of this kernel's time).
    // Two passes of 6 row tiles (96 rows) each: [96][BNW + 4] fp32 fits the operand ring.
    constexpr int CSTR = BNW + 4;                       // 4*CSTR = 16 (mod 64) banks: conflict-free writes
    static_assert(96 * CSTR * 4 <= 2 * STAGE, "epilogue tile must fit the operand ring");
    float* const ct = reinterpret_cast<float*>(lds);
    constexpr int NTHR = 128 * WN;
    constexpr int VPT = GLU ? 8 : 4;                    // output columns per thread per row
    constexpr int TPR = NOUT / VPT;                     // threads per output row
    constexpr int RPP = NTHR / TPR;                     // rows per sweep
    const int cq = (tid % TPR) * VPT;
    const bool vec_c = p.C && ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
    const bool vec_r = !p.resid || (((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.resid) & 15) == 0));
    const bool vec_s = p.C_hi && ((p.ldc_split & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.C_hi) & 15) == 0) &&
                       ((reinterpret_cast<uintptr_t>(p.C_lo) & 15) == 0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                // operand ring / previous half fully consumed
#pragma unroll
        for (int tt = 0; tt < PS_RT / 4; ++tt) {
            const int t = half * (PS_RT / 4) + tt;
            const int lr = (wm + 2 * tt) * 16 + (lane >> 4) * 4;   // row inside this half's 96-row tile
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int col;
                if (GLU) col = (c < 2) ? (wn * 32 + c * 16) : (NOUT + wn * 32 + (c - 2) * 16);
                else col = wn * 64 + c * 16;
#pragma unroll
                for (int r = 0; r < 4; ++r) ct[(lr + r) * CSTR + col + (lane & 15)] = acc[t][c][r];
            }
        }
        __syncthreads();
        for (int row = tid / TPR; row < 96; row += RPP) {
            const int m = m_begin + half * 96 + row;
            const int n = n_tile * NOUT + cq;
            if (m >= m_end || n >= p.N) continue;
            const long orow = p.c_rowidx ? p.c_rowidx[m] : m;
            float v[VPT];
            if (GLU) {
                const float* gp = &ct[row * CSTR + cq];
                const float* up = &ct[row * CSTR + NOUT + cq];
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) {
                    const float4 g = reinterpret_cast<const float4*>(gp)[q4];
                    const float4 u = reinterpret_cast<const float4*>(up)[q4];
                    v[q4 * 4 + 0] = silu_f(g.x) * u.x; v[q4 * 4 + 1] = silu_f(g.y) * u.y;
                    v[q4 * 4 + 2] = silu_f(g.z) * u.z; v[q4 * 4 + 3] = silu_f(g.w) * u.w;
                }
            } else {
                const float4 a4 = *reinterpret_cast<const float4*>(&ct[row * CSTR + cq]);
                v[0] = a4.x; v[1] = a4.y; v[2] = a4.z; v[3] = a4.w;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (n + q < p.N) {
                        float tv = v[q];
                        if (p.bias) tv += p.bias[n + q];
                        tv = apply_act(tv, p.act);
                        if (p.scale) tv *= p.scale[n + q];
                        v[q] = tv;
                    }
                }
                if (p.resid) {
                    if (vec_r && n + 3 < p.N) {
                        const float4 rr = *reinterpret_cast<const float4*>(p.resid + orow * p.ldr + n);
                        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (n + q < p.N) v[q] += p.resid[orow * p.ldr + n + q];
                    }
                }
            }
            const bool full = n + VPT - 1 < p.N;
            if (p.C) {
                if (vec_c && full) {
#pragma unroll
                    for (int q4 = 0; q4 < VPT / 4; ++q4)
                        reinterpret_cast<float4*>(p.C + orow * p.ldc + n)[q4] =
                            make_float4(v[q4 * 4], v[q4 * 4 + 1], v[q4 * 4 + 2], v[q4 * 4 + 3]);
                } else {
#pragma unroll
                    for (int q = 0; q < VPT; ++q) if (n + q < p.N) p.C[orow * p.ldc + n + q] = v[q];
                }
            }
            if (p.C_hi) {
                uint32_t hi[VPT], lo[VPT];
#pragma unroll
                for (int q = 0; q < VPT; ++q) split_bf16(v[q], hi[q], lo[q]);
                if (vec_s && full && VPT == 8) {
                    *reinterpret_cast<uint4*>(p.C_hi + orow * p.ldc_split + n) =
                        make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4 % VPT] | (hi[5 % VPT] << 16),
                                   hi[6 % VPT] | (hi[7 % VPT] << 16));
                    *reinterpret_cast<uint4*>(p.C_lo + orow * p.ldc_split + n) =
                        make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4 % VPT] | (lo[5 % VPT] << 16),
                                   lo[6 % VPT] | (lo[7 % VPT] << 16));
                } else {
#pragma unroll
                    for (int q = 0; q < VPT; ++q) {
                        if (n + q < p.N) {
                            p.C_hi[orow * p.ldc_split + n + q] = (uint16_t)hi[q];
                            p.C_lo[orow * p.ldc_split + n + q] = (uint16_t)lo[q];
                        }
                    }
                }
            }
        }
    }
}

// fp32 rows -> bf16 hi/lo planes (x = hi + lo to 2^-17)
__global__ void k_split_planes(const float* __restrict__ x, long ldx, uint16_t* __restrict__ hi,
                               uint16_t* __restrict__ lo, long ldo, int rows, int cols8) {
    const long total = (long)rows * cols8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols8;
        const int c = (int)(i - r * cols8);
        const float4 a = reinterpret_cast<const float4*>(x + r * ldx)[c * 2];
        const float4 b = reinterpret_cast<const float4*>(x + r * ldx)[c * 2 + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t h[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split_bf16(v[j], h[j], l[j]);
        reinterpret_cast<uint4*>(hi + r * ldo)[c] =
            make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        reinterpret_cast<uint4*>(lo + r * ldo)[c] =
            make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
    }
}

}  // namespace

int vhk_gemm_ps(hipStream_t st, const VhGemmPsArgs& a0) {
    VhGemmPsArgs a = a0;
    a.ablate = vh_tuning()->ps_ablate;   // timing experiments only (results are wrong when non-zero)
    if (a.K <= 0 || a.K % 64 != 0 || a.M < 0 || a.N <= 0 || (a.lda % 8) != 0 || (a.ldw % 8) != 0) return -1;
    if (!a.A_hi || !a.A_lo || !a.W || (!a.C && !a.C_hi) || (a.C_hi && !a.C_lo)) return -1;
    if (a.M == 0) return 0;
    const int slots = a.group_off ? (a.M / PS_BM + a.ngroups) : (a.M + PS_BM - 1) / PS_BM;
    if (a.W_up) {
        const int n_tiles = (a.N + 127) / 128;
        hipLaunchKernelGGL((k_gemm_ps<4, true>), dim3(n_tiles * slots), dim3(512), 0, st, a);
    } else if (a.wide) {
        const int n_tiles = (a.N + 255) / 256;
        hipLaunchKernelGGL((k_gemm_ps<4, false>), dim3(n_tiles * slots), dim3(512), 0, st, a);
    } else {
        const int n_tiles = (a.N + 127) / 128;
        hipLaunchKernelGGL((k_gemm_ps<2, false>), dim3(n_tiles * slots), dim3(256), 0, st, a);
    }
    return 0;
}

int vhk_split_planes(hipStream_t st, const float* x, long ldx, uint16_t* hi, uint16_t* lo, long ldo, int rows,
                     int cols) {
    if (cols % 8 != 0 || (ldx % 4) != 0 || (ldo % 8) != 0 || rows < 0) return -1;
    if (rows == 0) return 0;
    long total = (long)rows * (cols / 8);
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_split_planes, dim3((int)g), dim3(256), 0, st, x, ldx, hi, lo, ldo, rows, cols / 8);
    return 0;
}
