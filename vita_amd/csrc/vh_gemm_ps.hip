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

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((address_space(3))) void* lds_void_t;
typedef const __attribute__((address_space(1))) void* glb_void_t;

// timing experiments only (results are wrong when non-zero): 1 = no activation pieces, 2 = no weight pieces,
// 4 = no MFMAs, 8 = no fragment reads of the activations, 16 = no MFMAs in the weight-loading waves, 32 = no epilogue stores.  Built with -DPS_ABLATE=n by profiles/ablate_ps.sh.
#ifndef PS_ABLATE
#define PS_ABLATE 0
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
        if (!(PS_ABLATE & 2)) glds16<NTW>(base, o, w_dst0 + slot * MG_SLOT + i * 2 * MG_SUB);
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
        if (PS_ABLATE & 2) return;
        if (NTW) dst = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + o));
        else dst = *reinterpret_cast<const u32x4*>(base + o);
    };
    unsigned char* const wr_dst0 = lds + W_BASE + wu * 1024 + lane * 16;
    auto w_store = [&](int kt, int s16, const u32x4& v) __attribute__((always_inline)) {
        // stage kt lives in slots (2 kt) % 4, +1: sub-tiles 0-7 / 8-15
        if (!(PS_ABLATE & 2))
            *reinterpret_cast<u32x4*>(wr_dst0 + (((2 * kt) & 3) + (s16 >> 3)) * MG_SLOT + (s16 & 7) * MG_SUB) = v;
    };
    auto a_piece = [&](int kt, int i) __attribute__((always_inline)) {
        const int kc = kt < t.nk ? kt : t.nk - 1;
        const size_t kb = (size_t)(t.k0 + kc) * 128;
        const unsigned char* base = a_plane + kb;
        if (!(PS_ABLATE & 1)) glds16<false>(base, offa[i], a_dst0 + (kt & 1) * A_BUF + i * MG_SUB);
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
            if (s + 1 < NSTEP && !(PS_ABLATE & 8)) {
                const int ks1 = (s + 1) / (NSTEP / 2), rti1 = wm + 2 * ((s + 1) % (NSTEP / 2));
                nh_ = *reinterpret_cast<const bf16x8_t*>(ab + rti1 * MG_SUB + (ks1 ? fo1 : fo0));
                nl_ = *reinterpret_cast<const bf16x8_t*>(ab + (RTMAX + rti1) * MG_SUB + (ks1 ? fo1 : fo0));
            }
            if (RTE > 0 && !(PS_ABLATE & 4) && !((PS_ABLATE & 16) && MODE != 0)) {   // 16: the weight waves skip their MFMAs
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
        if ((PS_ABLATE & ~32) == 0) {
            __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);               // 8 weight + 2 activation fragment reads
            StepOrder<0, NSTEP, NP, FRONT>::pin();
            if (MODE == 3) __builtin_amdgcn_sched_group_barrier(0x200, 16, 0);   // the LDS stores
        }
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

    // ---- epilogue: acc[i][c][r] = out[token m_begin + (wm+2i)*16 + (lane&15)][col0(c) + 4*(lane>>4) + r] ----
    // The C^T accumulators give a lane 4 consecutive columns of ONE token: stored directly, a wave instruction touches 16
    // token rows x 64 B, rows tens of KB apart (r02).  For the K-split projections that is 40-70 MB of such stores per
    // launch, and it is where the slow boxes of the pool lose their time: QKV 133 us with the stores, 57 us without
    // (profiles/r03_proj_probe.txt; 70-80 us in all on a fast box).  So the tile goes through LDS (free after the K loop)
    // and leaves row by row: one wave instruction = 1 KB of ONE output row (fp32) / 256 B of two rows (bf16 planes).
    if (PS_ABLATE & 32) {                            // timing experiment: no output at all
#pragma unroll
        for (int i = 0; i < RTW; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) asm volatile("" ::"v"(acc[i][c]));
        return;
    }
    constexpr int NCOL = GLU ? 128 : 256;            // fp32 values per staged row
    constexpr int ROWB = NCOL * 4 + 16;              // + 16 B: consecutive rows start 4 banks apart
    constexpr int RC = GLU ? 192 : 96;               // rows per pass: 192 x 528 B = 99 KB / 96 x 1040 B = 97.5 KB
    static_assert(RC * ROWB <= 2 * A_BUF + NSLOT * MG_SLOT || RTMAX * 16 * ROWB <= 2 * A_BUF + NSLOT * MG_SLOT, "staging fits the rings");
    const int jrow = lane & 15, jc = (lane >> 4) * 4;
    const int tile_rows = t.m_end - t.m_begin;
    for (int r0 = 0; r0 < t.rt * 16; r0 += RC) {     // block-uniform
#pragma unroll
        for (int i = 0; i < RTW; ++i) {
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
            for (int rr = wid * 2; rr < nrows; rr += 16) {
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
            for (int row = wid; row < nrows; row += 8) {
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

template <bool GLU, int RTMAX, int NSLOT, bool NTW, bool WREG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_gemm_ps(const VhGemmPsArgs p) {
    constexpr int A_BUF = RTMAX * 2 * MG_SUB;
    constexpr int NOUT = GLU ? 128 : 256;
    static_assert(2 * A_BUF + NSLOT * MG_SLOT <= MG_LDS, "LDS budget");
    static_assert(NSLOT >= 4 && (RTMAX % 2) == 0 && (!WREG || NSLOT == 4), "ring geometry");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * A_BUF + NSLOT * MG_SLOT];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const int E = p.group_off ? p.ngroups : 1;
    const int NT = (p.N + NOUT - 1) / NOUT;
    const int nk_total = p.K >> 6;
    auto rows_of = [&](int e) { return p.group_off ? (p.group_off[e + 1] - p.group_off[e]) : p.M; };
    auto mtiles_of = [&](int rows) { return (((rows + 15) >> 4) + RTMAX - 1) / RTMAX; };

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
        int best = 1 << 30;
        for (int ks = 1; ks <= -p.ksplit && ks <= nk_total; ++ks) {
            const int Tx = (MT * NT * ks + 7) >> 3;                   // tiles of the fullest XCD
            const int Rr = Tx / nb, rr = Tx - Rr * nb;
            const int rounds16 = 16 * Rr + (rr == 0 ? 0 : (2 * rr <= nb ? 9 : 16));   // M-split last round ~ 0.55
            // + ~6 % of a full tile per round (prologue, epilogue) + what a slab costs to store and to sum again, in the
            // same units (a K = 4096 slab ~ 48; r03, profiles/r03_proj_variants.txt: O projection 5 slabs 57.5 us, 2 slabs 52.8)
            const int est = (rounds16 * 64) / ks + 4 * rounds16 + ks * ((48 * 64) / nk_total);
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
    const int g0 = (int)(((long)T * xcd) >> 3), g1 = (int)(((long)T * (xcd + 1)) >> 3);

    // Block j of the XCD takes tiles g0 + j, g0 + j + nb, ...  A run is seldom a multiple of nb (896 gate|up tiles
    // = 3.5 per CU): the tiles of the last, partial round are cut in TWO along M when that gives every block
    // something to do — both halves stream the same weight tile at the same time on the same XCD (second reader
    // hits L2), each multiplies half the rows; the full-width last round would leave half the CUs idle.
    const int Tx = g1 - g0;
    const int R = Tx / nb, r = Tx - R * nb;
    const bool split_tail = r > 0 && 2 * r <= nb;
    for (int it = 0; it <= R; ++it) {
        int g, half = -1;
        if (it < R) g = g0 + it * nb + j;
        else if (r == 0) break;
        else if (split_tail) { if (j >= 2 * r) break; g = g0 + R * nb + (j >> 1); half = j & 1; }
        else { if (j >= r) break; g = g0 + R * nb + j; }
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
    if (cfg < 0 || cfg > 1) cfg = avg <= 64 ? 0 : 1;
    int grid = num_cus();   // persistent: one 8-wave block per CU
    grid &= ~7;
    if (grid < 8) grid = 8;
    // non-temporal weight loads keep the activation planes in L2 (down projection: -7 %), but a run whose last
    // round is M-split relies on L2 for the second reader of each weight tile (gate|up: +5 % with nt)
    bool nt = vh_tuning()->ps_nt > 0;
    if (vh_tuning()->ps_nt < 0) {
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
