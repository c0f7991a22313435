// vh_gemm_ps.hip — the weight-streaming GEMM on PRE-SPLIT activations: Mixtral prefill MoE grouped GEMMs
// (HF MixtralExperts, modeling_mixtral.py:57-93 as reached from
// web_demo/vllm_tools/vllm_file/mixtral.py:405-422) and any other skinny-M contraction.
//
// Regime.  At prefill lengths of a few hundred tokens every expert sees ~S/4 rows (138 at S = 552): the job is
// a 1.88 GB (gate|up) / 0.94 GB (down) weight stream against a short M.  Round 1's two kernels were both
// bounded by WEIGHT BYTES IN FLIGHT: time = fabric bytes / min(HBM rate, in-flight W per CU x 256 / ~2 us):
// the 64x128 general kernel moved 2.8x the weights across the fabric, the first pre-split kernel kept one
// 16 KB weight stage in flight per CU (= 2 TB/s).  This kernel is built around that model:
//   * PERSISTENT grid, one 8-wave block per CU; block b serves XCD b % 8, and the tile list is cut into 8
//     contiguous runs (expert-major), so with 8 balanced experts XCD x streams expert x: the expert's
//     activation planes stay in that XCD's L2 and every weight byte crosses the fabric once;
//   * tile = ALL rows of the expert (up to RTMAX x 16) x 256 weight rows (GLU: 128 gate + 128 up rows ->
//     128 outputs): activation re-reads per weight byte are half those of a 128-row tile;
//   * operands go global -> LDS by LDS-DMA (global_load_lds, 16 B/lane, whole 128-B lines, no VGPR round
//     trip) into two rings: activations (hi + lo planes) double-buffered per BK = 64 stage, weights in a ring
//     of NSLOT 16-KB half-stage slots.  Loads are issued by ROLE: waves 0-3 only weights, waves 4-7 only
//     activations.  Vector loads retire in order per wave, so an L2-hit activation load queued behind an
//     HBM-miss weight load would wait for it; with the roles split the weight waves keep NSLOT-2 slots
//     (48 KB at NSLOT = 5) in flight behind a COUNTED s_waitcnt vmcnt(4*(NSLOT-4)) while the activation waves
//     run one stage ahead.  One raw s_barrier per stage;
//   * LDS image: 16-row x 128-B sub-tiles, each written lane-linearly by two LDS-DMA instructions (8 rows
//     each); the 16-B chunks of a row are permuted on the SOURCE side (chunk ^ (row >> 1)), which makes every
//     ds_read_b128 fragment read bank-conflict free (guide §5.4 rule 21);
//   * MFMA v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as the A operand and the activation fragment
//     as B, i.e. the accumulator holds C^T tiles: a lane owns 4 consecutive output columns of one token,
//     so the epilogue stores 16 B (fp32) / 8 B (bf16 planes) per lane without an LDS transpose.  hi and lo
//     activation planes accumulate into the same fp32 tile (exact mode, vh_common.h);
//   * waves 2 (M) x 4 (N): wave (wm, wn) owns row tiles wm, wm+2, ... and 64 weight rows (GLU: 32 gate +
//     32 up); waves w and w+4 share a SIMD, so each SIMD carries one wave of either M half (balanced when
//     the row-tile count is odd) and one loader of either role.  The K loop is instantiated per row-tile
//     count (RTW = 1 .. RTMAX/2), so there is no branch inside it;
//   * split-K (ksplit) for the down projection: N = 4096 gives only 128 tiles at 256 rows, so K is cut
//     in two and the partial sums go to separate output slabs that the combine kernel adds.
#include "vh_common.h"
#include "vh_kernels.h"

namespace {

#include "vh_gemm_ps_inl.h"

// One tile: prologue, K loop, epilogue.  RTW = row tiles of the waves with wm = 0 (the weight loaders), RTA = row tiles
// of the waves with wm = 1 (the activation loaders): RTW - 1 when the tile has an odd number of row tiles, so no
// wave multiplies a row tile that does not exist.
template <bool GLU, int RTMAX, int NSLOT, int RTW, int RTA, bool NTW, bool WREG>
__device__ __forceinline__ void run_tile(const VhGemmPsArgs& p, const TileCtx& t, unsigned char* lds, const int lane,
                                         const int wid) {
    constexpr int A_BUF = RTMAX * 2 * MG_SUB;    // one activation stage: hi sub-tiles then lo sub-tiles
    constexpr int W_BASE = 2 * A_BUF;
    constexpr int NOUT = GLU ? 128 : 256;
    constexpr int WAIT_W = 4 * (NSLOT - 4);      // weight LDS-DMAs that may stay in flight at a stage boundary
    const int wm = wid >> 2, wn = wid & 3;
    const int lrow = lane >> 3;
    const bool w_loader = wid < 4;               // wave-uniform (wid is an SGPR)

    // ---- per-lane source offsets of this wave's LDS-DMA pieces -----------------------------------------
    // weights: wave w owns pieces q = 16h + w + 4i of half-stage h: sub-tile (w>>1) + 2i (+8h), 8-row half w&1.
    // GLU: sub-tile s = 4*wn' + c holds gate rows (c < 2) or up rows (c >= 2) of output columns
    //      n0 + 32*wn' + 16*(c&1) ..; with s = (w>>1) + 2i (+8h) the piece is a gate piece for even i, up for odd i.
    // activations: wave w' = wid-4 loads plane w'>>1, 8-row half w'&1 of every row tile.
    // (separate named arrays with compile-time indices: a merged array indexed by the run-time half-stage
    // parity was placed in scratch memory by hipcc, i.e. scratch loads inside the K loop)
    uint32_t offw0[4], offw1[4], offa[RTMAX];
#pragma unroll
    for (int i = 0; i < 4; ++i) { offw0[i] = 0; offw1[i] = 0; }
#pragma unroll
    for (int i = 0; i < RTMAX; ++i) offa[i] = 0;
    if (w_loader) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = 8 * h + (wid >> 1) + 2 * i;
                const int r16 = (wid & 1) * 8 + lrow;
                int n;
                if (GLU) n = t.n0 + (s >> 2) * 32 + (s & 1) * 16 + r16;
                else n = t.n0 + s * 16 + r16;
                if (n > p.N - 1) n = p.N - 1;                       // clamped rows: products never stored
                const int c8 = (lane & 7) ^ ((r16 >> 1) & 7);
                const uint32_t o = (uint32_t)(((size_t)n * p.ldw + c8 * 8) * 2);
                if (h) offw1[i] = o; else offw0[i] = o;
            }
    } else {
        const int wl = wid - 4;
        const int r16 = (wl & 1) * 8 + lrow;
        const int c8 = (lane & 7) ^ ((r16 >> 1) & 7);
#pragma unroll
        for (int i = 0; i < RTMAX; ++i) {
            int m = t.m_begin + i * 16 + r16;
            if (m > t.m_end - 1) m = t.m_end - 1;
            const long src_row = p.a_rowidx ? p.a_rowidx[m] : m;
            offa[i] = (uint32_t)(((size_t)src_row * p.lda + c8 * 8) * 2);
        }
    }
    const unsigned char* a_plane = reinterpret_cast<const unsigned char*>((wid & 2) ? p.A_lo : p.A_hi);
    const unsigned char* w_gate = reinterpret_cast<const unsigned char*>(t.Wb);
    const unsigned char* w_up = reinterpret_cast<const unsigned char*>(GLU ? t.Wu : t.Wb);

    // fragment read offset inside a sub-tile for k-step ks: row r = lane&15, chunk = ks*4 + (lane>>4)
    const int fr = lane & 15;
    const int frag_base = (fr >> 3) * 1024 + (fr & 7) * 128;
    const int frag_x = (fr >> 1) & 7;
    const int fo0 = frag_base + (((lane >> 4)) ^ frag_x) * 16;
    const int fo1 = frag_base + ((4 + (lane >> 4)) ^ frag_x) * 16;

    f32x4 acc[RTW][4];
#pragma unroll
    for (int i = 0; i < RTW; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nh = 2 * t.nk;                      // weight half-stages of this tile
    constexpr int NPA = RTW + RTA;                // activation pieces per stage per A-loader = row tiles of the tile
    unsigned char* const w_dst0 = lds + W_BASE + (wid >> 1) * MG_SUB + (wid & 1) * 1024;
    unsigned char* const a_dst0 = lds + ((wid >> 1) & 1) * (RTMAX * MG_SUB) + (wid & 1) * 1024;

    // ---- the two ways a weight piece travels -----------------------------------------------------------
    // WREG = false: LDS-DMA into the NSLOT ring (lookahead NSLOT-2 half-stages; needs the LDS for it: small M).
    // WREG = true : global_load into staging registers TWO stages ahead, ds_write into a 2-stage LDS ring one
    //   stage ahead.  The lookahead lives in registers (64 VGPRs of the weight waves) instead of LDS: with all
    //   rows of an expert double-buffered (80-96 KB) only 4-5 DMA slots fit, i.e. ~1 stage of lead, and the
    //   stage time then converges to the HBM latency (measured 2.4 us per stage against 1.3 us of compute).
    //   The weight waves work in two PAIRS by stage parity: pair g loads stage s (s = g mod 2) during stage
    //   s-2 and stores it during stage s-1, and issues nothing else in between — so the vmcnt(0) hipcc puts in
    //   front of the stores waits for exactly those loads (with both stages' loads in ONE wave's in-order
    //   queue the needed wait is vmcnt(8), and hipcc emitted vmcnt(0) across the loop back edge: no lookahead).
    auto w_dma = [&](int hh, int slot, int i) __attribute__((always_inline)) {
        const int hc = hh < nh ? hh : nh - 1;     // clamped re-load at the tile's end lands in a slot nobody reads again
        const size_t kb = (size_t)(t.k0 + (hc >> 1)) * 128;
        const unsigned char* base = ((GLU && (i & 1)) ? w_up : w_gate) + kb;    // wave-uniform part: SGPR pair
        const uint32_t o = (hc & 1) ? offw1[i] : offw0[i];                      // per-lane part: one 32-bit VGPR
        glds16<NTW>(base, o, w_dst0 + slot * MG_SLOT + i * 2 * MG_SUB);
    };
    // register-staged: this wave owns the 8-row half u = wid&1 of ALL 16 sub-tiles of the stages of its parity
    const int wu = wid & 1, wg = (wid >> 1) & 1;
    const int wr16 = wu * 8 + lrow;
    const uint32_t wcol = (uint32_t)(((lane & 7) ^ ((wr16 >> 1) & 7)) * 16);
    auto w_load = [&](int kt, int s16, u32x4& dst) __attribute__((always_inline)) {   // sub-tile s16 of stage kt
        const int kc = kt < t.nk ? kt : t.nk - 1;
        const size_t kb = (size_t)(t.k0 + kc) * 128;
        const unsigned char* base = ((GLU && (s16 & 2)) ? w_up : w_gate) + kb;
        int n = GLU ? t.n0 + (s16 >> 2) * 32 + (s16 & 1) * 16 + wr16 : t.n0 + s16 * 16 + wr16;
        if (n > p.N - 1) n = p.N - 1;             // clamped rows: products never stored
        const uint32_t o = (uint32_t)n * (uint32_t)(p.ldw * 2) + wcol;
        if (NTW) dst = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + o));
        else dst = *reinterpret_cast<const u32x4*>(base + o);
    };
    unsigned char* const wr_dst0 = lds + W_BASE + wu * 1024 + lane * 16;
    auto w_store = [&](int kt, int s16, const u32x4& v) __attribute__((always_inline)) {
        // stage kt lives in slots (2 kt) % 4, +1: sub-tiles 0-7 / 8-15
        *reinterpret_cast<u32x4*>(wr_dst0 + (((2 * kt) & 3) + (s16 >> 3)) * MG_SLOT + (s16 & 7) * MG_SUB) = v;
    };
    auto a_piece = [&](int kt, int i) __attribute__((always_inline)) {
        const int kc = kt < t.nk ? kt : t.nk - 1;
        const size_t kb = (size_t)(t.k0 + kc) * 128;
        const unsigned char* base = a_plane + kb;
        glds16<false>(base, offa[i], a_dst0 + (kt & 1) * A_BUF + i * MG_SUB);
    };

    // ---- prologue --------------------------------------------------------------------------------------
    u32x4 wreg[16];                               // weight staging set (WREG)
#pragma unroll
    for (int q = 0; q < 16; ++q) wreg[q] = u32x4{0u, 0u, 0u, 0u};
    if (w_loader) {
        if (WREG) {
#pragma unroll
            for (int q = 0; q < 16; ++q) w_load(wg, q, wreg[q]);              // pair 0: stage 0, pair 1: stage 1
            if (wg == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) w_store(0, q, wreg[q]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NSLOT - 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) w_dma(s, s, i);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NPA; ++i) a_piece(0, i);
    }
    int slot_cur = 0;                             // ring slot of half-stage 2k
    int slot_fill = NSLOT - 2;                    // ring slot of half-stage 2k + NSLOT - 2 (filled during stage k)

    // One stage.  MODE 0: activation waves (LDS-DMA of stage k+1).  MODE 1: weight waves, DMA ring.  MODE 2 / 3:
    // register-staged weight waves in their LOAD stage (stage k+2 -> registers) / STORE stage (registers ->
    // LDS image of stage k+1).  The pieces are issued BETWEEN the MFMA groups (a piece costs ~60-100 issue
    // cycles), fragment reads run one step ahead of the MFMAs that consume them, the stores come last.  No
    // branch inside: every iteration issues the same number of pieces (clamped at the tile's end).
    auto stage = [&](const int k, auto mode_c) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_c)::value;
        constexpr int RTE = MODE == 0 ? RTA : RTW;                    // row tiles this wave multiplies
        constexpr int NSTEP = 2 * (RTE > 0 ? RTE : 1);                // (k-step, row tile) steps of 8 MFMAs per stage
        constexpr int NP = MODE == 0 ? NPA : (MODE == 1 ? 8 : (MODE == 2 ? 16 : 0));
        // loads go out in the first steps: a piece issued late in its stage has that much less time to land
        // before the wait at the stage boundary (activations: L2 hits one stage ahead; register-staged weights:
        // the stage time converges to HBM latency / minimum lead)
        constexpr int FRONT = (MODE == 2 || MODE == 0) ? (NP + (NSTEP >= 4 ? 3 : 1)) / (NSTEP >= 4 ? 4 : 2) : 0;
        const unsigned char* ab = lds + (k & 1) * A_BUF;
        const int slot1 = (slot_cur + 1 == NSLOT) ? 0 : slot_cur + 1;
        const int slot_w = (wn >> 1) ? slot1 : slot_cur;
        const unsigned char* wb = lds + W_BASE + slot_w * MG_SLOT + (wn & 1) * 4 * MG_SUB;
        const int fill1 = (slot_fill + 1 == NSLOT) ? 0 : slot_fill + 1;
        bf16x8_t bw[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int c = 0; c < 4; ++c) bw[ks][c] = *reinterpret_cast<const bf16x8_t*>(wb + c * MG_SUB + (ks ? fo1 : fo0));
        bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(ab + wm * MG_SUB + fo0);
        bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(ab + (RTMAX + wm) * MG_SUB + fo0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int ks = s / (NSTEP / 2), i = s % (NSTEP / 2);
#pragma unroll
            for (int q = piece_lo(s, NSTEP, NP, FRONT); q < piece_lo(s + 1, NSTEP, NP, FRONT); ++q) {   // this step's pieces
                if (MODE == 0) a_piece(k + 1, q);
                else if (MODE == 1) w_dma(2 * k + NSLOT - 2 + (q >> 2), (q >> 2) ? fill1 : slot_fill, q & 3);
                else if (MODE == 2) w_load(k + 2, q, wreg[q]);
            }
            bf16x8_t nh_ = ah, nl_ = al;
            if (s + 1 < NSTEP) {
                const int ks1 = (s + 1) / (NSTEP / 2), rti1 = wm + 2 * ((s + 1) % (NSTEP / 2));
                nh_ = *reinterpret_cast<const bf16x8_t*>(ab + rti1 * MG_SUB + (ks1 ? fo1 : fo0));
                nl_ = *reinterpret_cast<const bf16x8_t*>(ab + (RTMAX + rti1) * MG_SUB + (ks1 ? fo1 : fo0));
            }
            if (RTE > 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[ks][c], ah, acc[i][c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[ks][c], al, acc[i][c], 0, 0, 0);
            } else {
                asm volatile("" ::"v"(ah), "v"(al), "v"(bw[ks][0]), "v"(bw[ks][1]), "v"(bw[ks][2]), "v"(bw[ks][3]));
            }
            ah = nh_; al = nl_;
        }
        if (MODE == 3) {                          // stage k+1 (loaded during stage k-1) -> the LDS image stage k-1 used
#pragma unroll
            for (int q = 0; q < 16; ++q) w_store(k + 1, q, wreg[q]);
        }
        // pin that order: hipcc's scheduler otherwise sinks every fragment read to just before its MFMAs
        // (one register, lgkmcnt(0) in front of each group of four) and hoists all loads to the top
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);               // 8 weight + 2 activation fragment reads
        StepOrder<0, NSTEP, NP, FRONT>::pin();
        if (MODE == 3) __builtin_amdgcn_sched_group_barrier(0x200, 16, 0);   // the LDS stores
    };

    // the role branches are OUTSIDE the K loop (loops with matching barrier counts): with an if/else inside the
    // loop hipcc kept two copies of the accumulators (MFMAs with dst != src C, 256 VGPRs + spills)
    auto one = [&](int k, auto mode_c) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_c)::value;
        // stage k must be in LDS: the activation DMA of stage k / weight DMA slots <= 2k+1 (NSLOT-4 younger ones
        // may stay in flight) / this wave's LDS stores
        if (MODE == 0) wait_vm<0>();
        else if (MODE == 1) wait_vm<WAIT_W>();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // stage k visible to every wave; stage k-1 fully consumed
        __builtin_amdgcn_sched_barrier(0);
        stage(k, mode_c);
        __builtin_amdgcn_sched_barrier(0);
        slot_cur += 2;
        if (slot_cur >= NSLOT) slot_cur -= NSLOT;
        slot_fill += 2;
        if (slot_fill >= NSLOT) slot_fill -= NSLOT;
    };
    using M0 = std::integral_constant<int, 0>; using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>; using M3 = std::integral_constant<int, 3>;
    if (!w_loader) {
        for (int k = 0; k < t.nk; ++k) one(k, M0{});
    } else if (!WREG) {
        for (int k = 0; k < t.nk; ++k) one(k, M1{});
    } else if (wg == 0) {                         // pair 0: load on even stages, store on odd ones
        for (int k = 0; k < t.nk; k += 2) {
            one(k, M2{});
            if (k + 1 < t.nk) one(k + 1, M3{});
        }
    } else {                                      // pair 1: store on even stages, load on odd ones
        for (int k = 0; k < t.nk; k += 2) {
            one(k, M3{});
            if (k + 1 < t.nk) one(k + 1, M2{});
        }
    }
    wait_vm<0>();                                 // clamped tail pieces
    __builtin_amdgcn_s_barrier();                 // every wave is done with the rings before the next tile refills them

    tile_epilogue<GLU, RTMAX, RTW, 8, 2 * A_BUF + NSLOT * MG_SLOT, true>(p, t, lds, lane, wid, wm, wn, acc);
}

template <bool GLU, int RTMAX, int NSLOT, bool NTW, bool WREG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_gemm_ps(const VhGemmPsArgs p) {
    constexpr int A_BUF = RTMAX * 2 * MG_SUB;
    static_assert(2 * A_BUF + NSLOT * MG_SLOT <= MG_LDS, "LDS budget");
    static_assert(NSLOT >= 4 && (RTMAX % 2) == 0 && (!WREG || NSLOT == 4), "ring geometry");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * A_BUF + NSLOT * MG_SLOT];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    for_each_tile<GLU, RTMAX>(p, [&](const TileCtx& t) __attribute__((always_inline)) {
        // waves with wm = 0 take row tiles 0, 2, ..: ceil(rt / 2); waves with wm = 1 take 1, 3, ..: floor(rt / 2)
        switch (t.rt) {
#define PS_CASE(RT)                                                                                               \
    case RT:                                                                                                      \
        if constexpr (RTMAX >= RT) run_tile<GLU, RTMAX, NSLOT, (RT + 1) / 2, RT / 2, NTW, WREG>(p, t, lds, lane, wid); \
        break;
            PS_CASE(1) PS_CASE(2) PS_CASE(3) PS_CASE(4) PS_CASE(5) PS_CASE(6)
            PS_CASE(7) PS_CASE(8) PS_CASE(9) PS_CASE(10) PS_CASE(11) PS_CASE(12)
#undef PS_CASE
            default: break;
        }
    });
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
        uint32_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_bf16_pair(v[2 * j], v[2 * j + 1], h[j], l[j]);
        reinterpret_cast<uint4*>(hi + r * ldo)[c] = make_uint4(h[0], h[1], h[2], h[3]);
        reinterpret_cast<uint4*>(lo + r * ldo)[c] = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}

template <bool GLU, bool NTW>
int launch_cfg(hipStream_t st, const VhGemmPsArgs& a, int cfg, int grid) {
    // 0: <= 64 rows per m-tile, weights by LDS-DMA into an 8-slot ring (3 stages of lookahead fit next to the rows)
    // 1: <= 192 rows per m-tile, weights register-staged two stages ahead (default for the prefill MoE GEMMs)
    if (cfg == 0) hipLaunchKernelGGL((k_gemm_ps<GLU, 4, 8, NTW, false>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_gemm_ps<GLU, 12, 4, NTW, true>), dim3(grid), dim3(512), 0, st, a);
    return 0;
}

}  // namespace

int vhk_gemm_ps(hipStream_t st, const VhGemmPsArgs& a0) {
    VhGemmPsArgs a = a0;
    if (a.K <= 0 || a.K % 64 != 0 || a.M < 0 || a.N <= 0 || (a.lda % 8) != 0 || (a.ldw % 8) != 0) return -1;
    if (!a.A_hi || !a.A_lo || !a.W || (!a.C && !a.C_hi) || (a.C_hi && !a.C_lo)) return -1;
    if (a.ksplit == 0) a.ksplit = 1;
    if (a.ksplit < -8) return -1;                    // ksplit < 0: the kernel picks 1 .. -ksplit and reports it in *nslab_out
    if ((a.ksplit > 1 || a.ksplit < 0) && (a.W_up || a.bias || a.scale || a.resid || a.act != VH_ACT_NONE || a.C_hi || !a.C)) return -1;
    if (a.ksplit < 0 && !a.nslab_out) return -1;
    if (a.ksplit > (a.K >> 6)) return -1;
    if (a.group_off && a.ngroups > 8) return -1;     // the scheduler sorts at most 8 groups
    // 32-bit per-lane byte offsets inside one operand
    if ((size_t)a.ldw * (size_t)a.N * 2 >= (1ull << 32)) return -1;
    if (a.M == 0) return 0;
    // ring geometry by the expected rows per group (a group above RTMAX*16 rows is cut into balanced m-tiles)
    const int groups = a.group_off ? (a.ngroups > 0 ? a.ngroups : 1) : 1;
    const int avg = (a.M + groups - 1) / groups;
    int cfg = vh_tuning()->ps_cfg;
    // default: 64-row tiles with the weight DMA ring for small groups (batched decode iterations), else the 12-wave specialised
    // kernel (r04: 8 % faster than the 8-wave form on the MoE pair, 5-12 % on the projections: profiles/r04_sp_ab_*.txt, r04_proj_ab.txt)
    if (cfg < 0 || cfg > 2) cfg = avg <= 64 ? 0 : 2;
    int grid = num_cus();   // persistent: one 8-wave block per CU
    grid &= ~7;
    if (grid < 8) grid = 8;
    // Plain GEMMs with a host-known K split (the encoders' Linears, M ~ 250 .. 8000): a K = 1024 tile costs 25-30 us whatever its rows (16
    // stages of ~1.5 us + prologue / epilogue), so the launch is its number of ROUNDS: take the smallest m-tiles (>= 48 rows) whose tile
    // count still fits ONE round; problems that need several rounds anyway keep the 192-row tiles (fewest weight re-reads).
    // profiles/r04_enc_sp_sweep.jsonl: ViT qkv 34.3 -> 29.4 us (64-row tiles), fc1 36.0 -> 32.1 (80), 8-image batches unchanged.
    if (!a.group_off && a.ksplit >= 1 && a.rt_cap == 0) {
        const int nrt = (a.M + 15) >> 4, NT = (a.N + (a.W_up ? 127 : 255)) / (a.W_up ? 128 : 256);
        for (int c = 3; c < 12; ++c)
            if ((long)((nrt + c - 1) / c) * NT * a.ksplit <= grid) { a.rt_cap = c; break; }
    }
    // specialised waves (vh_gemm_sp.hip).  Weight loads WITHOUT the non-temporal hint unless forced (ps_nt = 1): same time (527 vs 530 us gate|up,
    // prefill 8.64-8.71 ms per 8 layers either way) and 6 % fewer fabric-side fetches (2.35 vs 2.49 GB: profiles/r04_fetch_nt_ab.txt)
    if (cfg == 2) {
        // auto = both (r06, profiles/r06_moe_xcd_ab.txt: down 258 -> 250 us, gate|up 505 -> 497 at uniform routing; 287 -> 286 / 537 -> 523 skewed)
        const int px = vh_tuning()->ps_xcd < 0 ? 3 : vh_tuning()->ps_xcd;
        // (grouped GEMMs only: the plain QKV / O projections measured 66.7 -> 70.6 and 52.2 -> 55.7 us with it, r06 calls 5 and 7)
        a.xcd_group = a.group_off ? ((a.W_up ? (px >> 1) : px) & 1) : 0;
        return vhk_gemm_sp(st, a, grid, vh_tuning()->ps_nt > 0);
    }
    // non-temporal weight loads keep the activation planes in L2 (down projection: -7 %), but a run whose last
    // round is M-split relies on L2 for the second reader of each weight tile (gate|up: +5 % with nt)
    bool nt = vh_tuning()->ps_nt > 0;
    if (vh_tuning()->ps_nt < 0 && a.group_off && avg <= 8) {
        // iterations of a few concurrent sequences: one row tile per expert, so no last-round tile is M-split and no weight tile has a second
        // reader (r05: 12.27 / 13.50 / 17.58 ms per iteration of 3 / 4 / 8 sequences with the hint against 12.61 / 13.90 / 18.19 without)
        nt = true;
    } else if (vh_tuning()->ps_nt < 0) {
        const int NOUT = a.W_up ? 128 : 256;
        const long T = (long)groups * ((avg + (cfg == 0 ? 63 : 191)) / (cfg == 0 ? 64 : 192)) * ((a.N + NOUT - 1) / NOUT) * (a.ksplit > 0 ? a.ksplit : 2);
        const long nb = grid / 8, Tx = (T + 7) / 8, r = Tx % nb;
        nt = !(r > 0 && 2 * r <= nb);
    }
    if (a.W_up) return nt ? launch_cfg<true, true>(st, a, cfg, grid) : launch_cfg<true, false>(st, a, cfg, grid);
    return nt ? launch_cfg<false, true>(st, a, cfg, grid) : launch_cfg<false, false>(st, a, cfg, grid);
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
