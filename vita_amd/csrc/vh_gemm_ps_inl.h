// vh_gemm_ps_inl.h — pieces shared by the two weight-streaming GEMM kernels on pre-split activations
// (vh_gemm_ps.hip: every wave loads and multiplies; vh_gemm_sp.hip: loader waves and MFMA waves): LDS image, LDS-DMA,
// tile list + persistent schedule, epilogue through LDS.  Included inside each file's anonymous namespace.
#pragma once

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((address_space(3))) void* lds_void_t;
typedef const __attribute__((address_space(1))) void* glb_void_t;

// (The compile-time ablation switches PS_ABLATE / FA_ABLATE of r02-r04 — one component compiled out per build, results in
// profiles/EXPERIMENTS.md — were removed in r05; the scripts that built them are in profiles/scripts/ablate/ and need the r04 tree.)
#ifndef PS_SCHED
#define PS_SCHED 0                  // tile -> block schedule of the including kernel: 0 = contiguous run per XCD (k_gemm_ps), 1 = cost-ordered grid stride (k_gemm_sp sets it)
#endif

#define MG_SUB 2048                  // one 16-row x 128-byte (BK = 64) sub-tile
#define MG_SLOT 16384                // weight half-stage slot: 8 sub-tiles = 128 weight rows
#define MG_LDS 163840                // 160 KiB

// LDS-DMA of 16 B per lane: lane l's bytes land at lds_dst + 16*l (wave-uniform LDS byte address, through M0);
// the source is base (wave-uniform, SGPR pair) + off (per lane, 32-bit).  INLINE ASM on purpose: with the
// __builtin_amdgcn_global_load_lds form hipcc (ROCm 7.2) turns every later `s_waitcnt lgkmcnt(N)` of the wave into
// lgkmcnt(0) — the fragment reads issued two steps ahead were waited for at once, and the matrix pipe idled
// ~40 % of each stage (both waves of a SIMD parked on LDS at the same time).  hipcc does not count this load:
// the waves that issue it wait with their own counted vmcnt (wait_vm) and never mix it with ordinary loads.
// M0 is saved and restored inside the statement (guide 5.7: the compiler owns M0).
template <bool NT>
__device__ __forceinline__ void glds16(const unsigned char* base, uint32_t off, unsigned char* lds_dst) {
    const uint32_t dst = (uint32_t)(uintptr_t)(lds_void_t)lds_dst;
    uint32_t keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// The instruction order of one stage, step by step (sched_group_barrier wants literal arguments):
//   [this step's LDS-DMA pieces] [the next step's two fragment reads] [this step's 8 MFMAs]
// pieces [piece_lo(S), piece_lo(S+1)) are issued in step S.  FRONT > 0: FRONT pieces per step from step 0 (the
// register-staged weight loads: a piece issued late in its stage has one stage less of lead, and the stage
// time converges to (HBM latency) / (minimum lead in stages)); else spread evenly.
__host__ __device__ constexpr int piece_lo(int S, int NSTEP, int NP, int FRONT) {
    return FRONT > 0 ? (S * FRONT < NP ? S * FRONT : NP) : (S * NP) / NSTEP;
}
template <int S, int NSTEP, int NP, int FRONT>
struct StepOrder {
    static __device__ __forceinline__ void pin() {
        constexpr int npc = piece_lo(S + 1, NSTEP, NP, FRONT) - piece_lo(S, NSTEP, NP, FRONT);
        if constexpr (npc > 0) __builtin_amdgcn_sched_group_barrier(0x020, npc, 0);
        if constexpr (S + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        if constexpr (S + 1 < NSTEP) StepOrder<S + 1, NSTEP, NP, FRONT>::pin();
    }
};

struct TileCtx {
    // tile coordinates (block-uniform)
    int m_begin, m_end, rt;          // activation rows [m_begin, m_end), rt = row tiles holding data
    int n0;                          // first output column of the tile
    int k0, nk;                      // first K stage and stage count of this K split
    int ks;                          // K split index
    const uint16_t* Wb; const uint16_t* Wu;
};


// ---- epilogue: acc[i][c][r] = out[token m_begin + (wm+2i)*16 + (lane&15)][col0(c) + 4*(lane>>4) + r] ----------------------
// NW = waves of the block (all of them store rows), HAS_ACC: the calling wave holds accumulators (the loader waves of the
// specialised kernel do not: they only take part in the barriers and the row stores), LDS_BYTES = the block's LDS (the rings are
// free after the K loop: the caller has passed a barrier behind the last fragment read).
template <bool GLU, int RTMAX, int RTW, int NW, int LDS_BYTES, bool HAS_ACC>
__device__ __forceinline__ void tile_epilogue(const VhGemmPsArgs& p, const TileCtx& t, unsigned char* lds, const int lane,
                                              const int wid, const int wm, const int wn, f32x4 (&acc)[RTW][4]) {
    // The C^T accumulators give a lane 4 consecutive columns of ONE token: stored directly, a wave instruction touches 16
    // token rows x 64 B, rows tens of KB apart (r02).  For the K-split projections that is 40-70 MB of such stores per
    // launch, and it is where the slow boxes of the pool lose their time: QKV 133 us with the stores, 57 us without
    // (profiles/r03_proj_probe.txt; 70-80 us in all on a fast box).  So the tile goes through LDS (free after the K loop)
    // and leaves row by row: one wave instruction = 1 KB of ONE output row (fp32) / 256 B of two rows (bf16 planes).
    constexpr int NCOL = GLU ? 128 : 256;            // fp32 values per staged row
    constexpr int ROWB = NCOL * 4 + 16;              // + 16 B: consecutive rows start 4 banks apart
    constexpr int RC = GLU ? 192 : 96;               // rows per pass: 192 x 528 B = 99 KB / 96 x 1040 B = 97.5 KB
    static_assert(RC * ROWB <= LDS_BYTES || RTMAX * 16 * ROWB <= LDS_BYTES, "staging fits the rings");
    const int jrow = lane & 15, jc = (lane >> 4) * 4;
    const int tile_rows = t.m_end - t.m_begin;
    for (int r0 = 0; r0 < t.rt * 16; r0 += RC) {     // block-uniform
#pragma unroll
        for (int i = 0; i < RTW; ++i) {
            if constexpr (!HAS_ACC) break;
            const int rl = (wm + 2 * i) * 16 - r0;   // first local row of this row tile (wave-uniform)
            if (rl < 0 || rl + 16 > RC || rl + 16 > RTMAX * 16) continue;
            unsigned char* rowp = lds + (size_t)(rl + jrow) * ROWB;
            if (GLU) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = silu_f(acc[i][c][r]) * acc[i][c + 2][r];
                    *reinterpret_cast<f32x4*>(rowp + (wn * 32 + c * 16 + jc) * 4) = v;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(rowp + (wn * 64 + c * 16 + jc) * 4) = acc[i][c];
            }
        }
        __syncthreads();
        const int nrows = min(RC, tile_rows - r0);
        if (GLU) {
            // two rows per wave instruction: lane -> row (lane >> 5), columns 4 (lane & 31) ..
            const int n = t.n0 + (lane & 31) * 4;
            for (int rr = wid * 2; rr < nrows; rr += 2 * NW) {
                const int row = rr + (lane >> 5);
                if (row >= nrows || n >= p.N) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(lds + (size_t)row * ROWB + (lane & 31) * 16);
                const int m = t.m_begin + r0 + row;
                const long orow = p.c_rowidx ? p.c_rowidx[m] : m;
                const bool full = n + 3 < p.N;
                if (p.C) {
                    float* cp = p.C + orow * p.ldc + n;
                    if (full && ((p.ldc & 3) == 0)) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                    else
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) cp[r] = v[r];
                }
                if (p.C_hi) {
                    uint32_t hi[2], lo[2];
                    split_bf16_pair(v[0], v[1], hi[0], lo[0]);
                    split_bf16_pair(v[2], v[3], hi[1], lo[1]);
                    uint16_t* hp = p.C_hi + orow * p.ldc_split + n;
                    uint16_t* lp = p.C_lo + orow * p.ldc_split + n;
                    if (full && ((p.ldc_split & 3) == 0)) {
                        *reinterpret_cast<uint2*>(hp) = make_uint2(hi[0], hi[1]);
                        *reinterpret_cast<uint2*>(lp) = make_uint2(lo[0], lo[1]);
                    } else {
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) { hp[r] = (uint16_t)(hi[r >> 1] >> (16 * (r & 1))); lp[r] = (uint16_t)(lo[r >> 1] >> (16 * (r & 1))); }
                    }
                }
            }
        } else {
            // one row per wave instruction: lane -> columns 4 lane ..
            const int n = t.n0 + lane * 4;
            for (int row = wid; row < nrows; row += NW) {
                if (n >= p.N) continue;
                const f32x4 a = *reinterpret_cast<const f32x4*>(lds + (size_t)row * ROWB + lane * 16);
                const int m = t.m_begin + r0 + row;
                const long orow = p.c_rowidx ? p.c_rowidx[m] : m;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float tv = a[r];
                    if (n + r < p.N) {
                        if (p.bias) tv += p.bias[n + r];
                        tv = apply_act(tv, p.act);
                        if (p.scale) tv *= p.scale[n + r];
                        if (p.resid) tv += p.resid[orow * p.ldr + n + r];
                    }
                    v[r] = tv;
                }
                const bool full = n + 3 < p.N;
                if (p.C) {
                    float* cp = p.C + (size_t)t.ks * p.c_split_stride + orow * p.ldc + n;
                    if (full && ((p.ldc & 3) == 0) && ((p.c_split_stride & 3) == 0))
                        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                    else
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) cp[r] = v[r];
                }
                if (p.C_hi) {
                    uint32_t hi[2], lo[2];
                    split_bf16_pair(v[0], v[1], hi[0], lo[0]);
                    split_bf16_pair(v[2], v[3], hi[1], lo[1]);
                    uint16_t* hp = p.C_hi + orow * p.ldc_split + n;
                    uint16_t* lp = p.C_lo + orow * p.ldc_split + n;
                    if (full && ((p.ldc_split & 3) == 0)) {
                        *reinterpret_cast<uint2*>(hp) = make_uint2(hi[0], hi[1]);
                        *reinterpret_cast<uint2*>(lp) = make_uint2(lo[0], lo[1]);
                    } else {
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) { hp[r] = (uint16_t)(hi[r >> 1] >> (16 * (r & 1))); lp[r] = (uint16_t)(lo[r >> 1] >> (16 * (r & 1))); }
                    }
                }
            }
        }
        __syncthreads();                             // the staging rows are free again (next pass / next tile's rings)
    }
}

// ---- tile list and persistent schedule: calls f(const TileCtx&) for every tile of this block, in order ---------------------
template <bool GLU, int RTMAX, class F>
__device__ __forceinline__ void for_each_tile(const VhGemmPsArgs& p, F&& f) {
    constexpr int NOUT = GLU ? 128 : 256;
    const int E = p.group_off ? p.ngroups : 1;
    const int NT = (p.N + NOUT - 1) / NOUT;
    const int nk_total = p.K >> 6;
    auto rows_of = [&](int e) { return p.group_off ? (p.group_off[e + 1] - p.group_off[e]) : p.M; };
    const int cap = (p.rt_cap > 0 && p.rt_cap < RTMAX) ? p.rt_cap : RTMAX;
    auto mtiles_of = [&](int rows) { return (((rows + 15) >> 4) + cap - 1) / cap; };

    // ---- tile list and its partition ---------------------------------------------------------------------
    int ord[8], n_exp = 0;
    for (int e = 0; e < E && e < 8; ++e) ord[n_exp++] = e;
    for (int i = 1; i < n_exp; ++i)                                   // insertion sort by rows, descending (<= 8 entries)
        for (int k = i; k > 0 && rows_of(ord[k]) > rows_of(ord[k - 1]); --k) { const int tmp = ord[k]; ord[k] = ord[k - 1]; ord[k - 1] = tmp; }
    int MT = 0;                                                       // m-tiles over all experts
    for (int i = 0; i < n_exp; ++i) MT += mtiles_of(rows_of(ord[i]));
    const int nb = gridDim.x >> 3;
    // K split chosen HERE when the caller allows it (p.ksplit < 0: up to -p.ksplit): with ~1 tile per CU the
    // makespan is set by the rounding of tiles / CUs — 256 tiles at 8 balanced experts, 320 when two experts need
    // two m-tiles (1.25 rounds at ks = 2, but 1.875 rounds of 2/3-size tiles at ks = 3) — so every block evaluates
    // rounds(ks) / ks + a per-tile overhead and takes the minimum; the count goes to *nslab_out for the reducer.
    int KS = p.ksplit > 1 ? p.ksplit : 1;
    if (p.ksplit < 0) {
        // Iterations of a few concurrent sequences (every expert <= 16 rows: ONE row tile per m-tile): a last-round tile cannot be cut along M
        // (its second half would be empty), so a partial round costs a whole one; and a slab is a few rows, not a prefill's hundreds — its price
        // scales with the rows.  (r05, profiles/r05_kernel_stats_concurrent_b3.txt: with the M-split credit and the prefill's slab price, 4
        // touched experts took ks = 2 — 128 tiles on half the CUs — where ks = 4 fills the chip: B = 3 at 12.6 ms per iteration against r02's 11.9.)
        // (8-wave kernels only: the specialised kernel takes tall tiles, and its code stays what the r05 prefill numbers were measured on)
#if PS_SCHED == 0
        const bool one_rt = n_exp > 0 && rows_of(ord[0]) <= 16;
        const int slab_rows = p.M < 512 ? p.M : 512;
#else
        constexpr bool one_rt = false;
        constexpr int slab_rows = 512;
#endif
        int best = 1 << 30;
        for (int ks = 1; ks <= -p.ksplit && ks <= nk_total; ++ks) {
            const int Tx = (MT * NT * ks + 7) >> 3;                   // tiles of the fullest XCD
            const int Rr = Tx / nb, rr = Tx - Rr * nb;
            const int rounds16 = 16 * Rr + (rr == 0 ? 0 : ((2 * rr <= nb && !one_rt) ? 9 : 16));   // M-split last round ~ 0.55
            // + ~6 % of a full tile per round (prologue, epilogue) + what a slab costs to store and to sum again, in the
            // same units (a K = 4096 slab of >= 512 rows ~ 48; r03, profiles/r03_proj_variants.txt: O projection 5 slabs 57.5 us, 2 slabs 52.8)
            const int est = (rounds16 * 64) / ks + 4 * rounds16 + ks * (((48 * 64) / nk_total) * slab_rows >> 9);
            if (est < best) { best = est; KS = ks; }
        }
        if (p.nslab_out && blockIdx.x == 0 && threadIdx.x == 0) *p.nslab_out = KS;
    }
    const int T = MT * NT * KS;
    // XCD x serves the contiguous run [T x / 8, T (x+1) / 8) of the tile list: experts by DECREASING row count (real
    // routers are far from uniform: 76 .. 307 rows per expert at S = 568 on the synthetic model), tiles expert-major:
    // an XCD streams 1-3 experts whose activation planes stay in its L2, and block j takes tiles j, j + nb, ... of the
    // run — the cheapest tiles in its last round.  (Runs of equal COST instead of equal count were tried: with ~1 tile
    // per CU an uneven count costs a whole extra round: down projection 264 -> 426 us.)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
#if PS_SCHED == 1
    // (r04 experiment) COST-ORDERED GRID STRIDE: round `it` is the 8 nb consecutive tiles [it * 8 nb, (it + 1) * 8 nb) of the list
    // (experts by decreasing rows), one per block: the tiles of a round cost about the same, every block gets one tile of every
    // cost class, and all XCDs work on the same 2-3 experts at a time.  With the contiguous per-XCD runs below XCD 0 holds only
    // the LARGEST experts' tiles and XCD 7 the smallest: the launch lasts as long as XCD 0.  Position q of a block inside a round
    // keeps the pairs (q, q ^ 1) — the two m-tiles of one weight tile, or the two halves of an M-split tail tile — on one XCD.
    const int stride = 8 * nb;
    // (r04: giving XCD x the contiguous chunk [x nb, (x + 1) nb) of every round instead — one expert's activation planes per XCD
    // per round — changed neither the time (583 vs 584 us uniform, 635-649 vs 634-636 skewed) nor FETCH_SIZE (2.436 vs 2.431 GB per
    // launch): profiles/r04_sched_map_ab.txt.)
    // (r06) p.xcd_group: XCD x takes the CONTIGUOUS positions [x nb, (x + 1) nb) of the round instead — with the list expert-major, K slice,
    // n-tile, m-tile fastest, these are the n-tiles of one or two (expert, K slice) groups: they stream the SAME activation rows stage by
    // stage, so a line of the h planes (63 MB at S = 552: far beyond the L2s) is filled into ONE L2 and hit by the group's other tiles,
    // where the interleaved placement has every XCD fetch every group's rows.
    const int jj = (nb & 1) ? (int)blockIdx.x : (p.xcd_group ? (xcd * nb + j) : (((j >> 1) << 4) | (xcd << 1) | (j & 1)));
    const int g0 = 0, Tx = T;
#else
    const int g0 = (int)(((long)T * xcd) >> 3), g1 = (int)(((long)T * (xcd + 1)) >> 3);

    // Block j of the XCD takes tiles g0 + j, g0 + j + nb, ...  A run is seldom a multiple of nb (896 gate|up tiles
    // = 3.5 per CU): the tiles of the last, partial round are cut in TWO along M when that gives every block
    // something to do — both halves stream the same weight tile at the same time on the same XCD (second reader
    // hits L2), each multiplies half the rows; the full-width last round would leave half the CUs idle.
    const int Tx = g1 - g0, stride = nb, jj = j;
#endif
    const int R = Tx / stride, r = Tx - R * stride;
    const bool split_tail = r > 0 && 2 * r <= stride;
    for (int it = 0; it <= R; ++it) {
        int g, half = -1;
        if (it < R) g = g0 + it * stride + jj;
        else if (r == 0) break;
        else if (split_tail) { if (jj >= 2 * r) break; g = g0 + R * stride + (jj >> 1); half = jj & 1; }
        else { if (jj >= r) break; g = g0 + R * stride + jj; }
        // ---- decode tile g: expert (in sorted order), then (ks, n-tile, m-tile) with the m-tile fastest ----------
        int e = 0, li = g, rows = 0, mt = 0, oi = 0;
        for (; oi < n_exp; ++oi) {
            e = ord[oi];
            rows = rows_of(e);
            mt = mtiles_of(rows);
            const int cnt = mt * NT * KS;
            if (li < cnt) break;
            li -= cnt;
        }
        if (oi == n_exp) break;
        const int mi = li % mt;
        li /= mt;
        const int nt = li % NT, ks = li / NT;
        const int nrt = (rows + 15) >> 4;
        const int rtper = (nrt + mt - 1) / mt;                      // balanced m-tiles
        const int off_e = p.group_off ? p.group_off[e] : 0;
        TileCtx t;
        t.m_begin = off_e + mi * rtper * 16;
        t.m_end = min(off_e + rows, t.m_begin + rtper * 16);
        if (t.m_begin >= t.m_end) continue;                         // (possible only with unbalanced remainders)
        t.rt = (t.m_end - t.m_begin + 15) >> 4;
        if (half >= 0) {                                            // M-split of a last-round tile
            const int h0 = (t.rt + 1) >> 1;
            if (half == 0) t.m_end = min(t.m_end, t.m_begin + h0 * 16);
            else t.m_begin += h0 * 16;
            if (t.m_begin >= t.m_end) continue;
            t.rt = (t.m_end - t.m_begin + 15) >> 4;
        }
        t.n0 = nt * NOUT;
        t.ks = ks;
        t.k0 = (int)(((long)nk_total * ks) / KS);
        t.nk = (int)(((long)nk_total * (ks + 1)) / KS) - t.k0;
        t.Wb = p.W + (size_t)e * p.w_group_stride;
        t.Wu = GLU ? p.W_up + (size_t)e * p.w_group_stride : nullptr;
        f(t);
    }
}
