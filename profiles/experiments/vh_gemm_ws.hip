// vh_gemm_ws.hip — weight-streaming GEMM on pre-split activations, round-3 form: the weights never touch LDS.
// Same contract as vh_gemm_ps.hip (VhGemmPsArgs): C[m, n] = epilogue(sum_k (A_hi + A_lo)[m, k] * W[n, k]); used for the
// Mixtral prefill MoE grouped GEMMs (HF MixtralExperts as reached from web_demo/vllm_tools/vllm_file/mixtral.py:405-422),
// the QKV / O projections and the batched-decode MoE.
//
// Why a second kernel.  r02's k_gemm_ps staged the 32 KB weight slab of a K stage through registers into LDS and read it
// back as fragments: 510 us of load phase + 334 us of matrix phase per gate|up launch that ADD UP (594-690 us measured,
// HBM floor 235 us).  Its two-slot LDS rings gave every load ONE stage of lead, so a stage could not be shorter than a
// memory round trip, and the weight path cost an LDS write + an LDS read per byte next to the activation traffic.
// Here:
//   * a wave owns 32 weight rows (GLU: 16 gate + 16 up rows of the same 16 output columns) and ALL row tiles of the
//     m-tile (waves 1 (M) x 8 (N)): its weight fragments are consumed by nobody else, so they are loaded from global
//     memory STRAIGHT INTO REGISTERS in the MFMA A-operand layout (lane = row (l & 15), 16-byte k-chunk (l >> 4); a
//     wave-instruction reads 16 rows x 64 B, the two k-steps of a stage cover whole 128-B lines back to back) into a
//     ring of NW = NS + 1 register sets of 16 VGPRs (v[192:255], see ws_ldw): NS stages (3-4 us) of lead, no LDS traffic,
//     no barrier on the path;
//   * the activation planes (shared by the 8 waves) come by LDS-DMA into a ring of NS = 2..4 stage slots (rt x 4 KB
//     each, as many as fit 160 KB): NS - 1 stages of lead.  One raw s_barrier per K stage;
//   * every load is inline asm and every wait a COUNTED s_waitcnt vmcnt(N): per stage a wave issues first its LDS-DMA
//     pieces of stage k + NS - 1, then its weight loads of stage k + NS, so the wait for "my pieces of stage k have
//     landed" (the oldest thing the barrier needs) never forces anything younger than the weights of stage k itself;
//   * the tile is all rows of the expert (up to 14 row tiles = 224 rows: accumulators 8 x rt VGPRs) x 256 weight
//     rows, as in r02; scheduling (experts by decreasing rows, one XCD per run of the tile list, M-split last round,
//     device-chosen K split) is r02's.
// MFMA operands: weight fragment = A, activation fragment = B, so the accumulators hold C^T (a lane owns 4 consecutive
// output columns of one token) and the epilogue needs no cross-lane traffic.  hi and lo planes accumulate into the same
// fp32 tile (exact mode, vh_common.h).
#include "vh_common.h"
#include "vh_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((address_space(3))) void* lds_void_t;

// timing experiments only (results are wrong when non-zero): 1 = no activation DMA, 2 = no weight loads, 4 = no MFMAs,
// 8 = only the hi plane's DMA pieces (half the activation traffic)
#ifndef WS_ABLATE
#define WS_ABLATE 0
#endif

#ifndef WS_NT
#define WS_NT 1                      // non-temporal weight loads (every weight byte is read once)
#endif
#ifndef WS_FD
#define WS_FD 2                      // activation fragment sets in flight per wave (fragment reads run WS_FD - 1 steps ahead)
#endif
#define WS_SUB 2048                  // one 16-row x 128-byte (BK = 64) sub-tile
#define WS_LDS 163840                // 160 KiB
#define WS_RTMAX 14                  // accumulators 8 x rt: 15 row tiles no longer fit below the ring without spilling

#ifndef WS_NSMAX
#define WS_NSMAX 3                   // at most 3 activation slots (2 stages of lead)
#endif
#ifndef WS_PWX
#define WS_PWX 1                     // weight loads lead the activation DMA by this many stages (0 or 1); sets = lead + 1 <= 4
#endif
__host__ __device__ constexpr int ws_slots(int RT) { return (WS_LDS / (RT * 2 * WS_SUB)) > WS_NSMAX ? WS_NSMAX : (WS_LDS / (RT * 2 * WS_SUB)); }

// LDS-DMA of 16 B per lane (see vh_gemm_ps.hip: inline asm so that hipcc's lgkmcnt bookkeeping of the fragment reads
// stays counted; M0 saved and restored inside the statement).
__device__ __forceinline__ void ws_glds16(const unsigned char* base, uint32_t off, unsigned char* lds_dst) {
    // (readfirstlane: the address IS wave-uniform, but in the 18-way instantiated kernel hipcc no longer proves it)
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t)lds_dst);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
}

// ---- the weight ring: v[WS_RING0 : 255], named LITERALLY ------------------------------------------------------------
// Weight fragment load: 16 B per lane from base (wave-uniform, SGPR pair) + off (per lane) + IMM into v[R0 : R0 + 3].
// The four sets of 16 registers at the top of the 256-register budget are written asynchronously by these loads and
// read only by ws_rdw() behind a hand-counted wait; no C++ value ever lives in them, so hipcc has nothing to copy, move
// or spill while a load is in flight.  The kernel caps hipcc's own allocation at v0..v191 (amdgpu_num_vgpr, see
// k_gemm_ws); that it never touches the ring and does not spill is CHECKED at build time on the emitted ISA
// (profiles/audit_ws.py, run by __graft_entry__.build(): no instruction outside these asm statements may name
// v[WS_RING0..255], no scratch).
// What was tried first and why it is not here (hipcc / ROCm 7.2):
//   * "+v" asm loads with a "+v" counted-wait statement: the copy for the wait's tied operands was emitted BEFORE the
//     wait, i.e. of registers still in flight (memory faults on the GPU);
//   * a ring of accumulation registers a[0:63]: any AGPR in asm text halves the VGPR budget to 128 and the allocator then
//     spills INTO those AGPRs; amdgpu_num_vgpr and dead "={a[..]}" outputs do not change the split;
//   * ordinary loads that hipcc counts: its waits become vmcnt(0) as soon as the loads sit behind the tail guards or
//     cross the K loop's back edge — every stage drained the whole queue.
#define WS_RING0 192
template <bool NT, int R0, int IMM>
__device__ __forceinline__ void ws_ldw(const unsigned char* base, uint32_t off) {
    static_assert(R0 >= WS_RING0 && R0 + 3 <= 255, "ring register");
    if (WS_ABLATE & 2) return;
    if (NT) asm volatile("global_load_dwordx4 v[%c2:%c3], %0, %1 offset:%c4 nt" ::"v"(off), "s"(base), "i"(R0), "i"(R0 + 3), "i"(IMM) : "memory");
    else asm volatile("global_load_dwordx4 v[%c2:%c3], %0, %1 offset:%c4" ::"v"(off), "s"(base), "i"(R0), "i"(R0 + 3), "i"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void ws_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// v[R0 : R0 + 3] -> one MFMA operand in compiler-owned VGPRs (volatile: stays behind the counted wait; the trailing nop
// covers VALU write -> MFMA read, which hipcc does not pad for an asm statement)
template <int R0>
__device__ __forceinline__ bf16x8_t ws_rdw() {
    u32x4 r;
    asm volatile("v_mov_b32 %0, v%c4\n\tv_mov_b32 %1, v%c5\n\tv_mov_b32 %2, v%c6\n\tv_mov_b32 %3, v%c7\n\ts_nop 1"
                 : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : "i"(R0), "i"(R0 + 1), "i"(R0 + 2), "i"(R0 + 3));
    bf16x8_t o;
    __builtin_memcpy(&o, &r, 16);
    return o;
}

// a pointer the compiler cannot prove wave-uniform (it depends on the tile list walked above) -> SGPR pair
__device__ __forceinline__ const unsigned char* ws_uniform(const void* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const unsigned char*>((uintptr_t)(((uint64_t)hi << 32) | lo));
}

// f(integral_constant<0>) ... f(integral_constant<N - 1>): an unrolled loop the compiler cannot re-roll
template <int I, int N, typename F>
__device__ __forceinline__ void ws_static_for(F& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ws_static_for<I + 1, N>(f);
    }
}

struct WsTile {
    int m_begin, m_end, rt;          // activation rows [m_begin, m_end), rt = row tiles holding data
    int n0;                          // first output column of the tile
    int k0, nk;                      // first K stage and stage count of this K split
    int ks;                          // K split index
    const uint16_t* Wb; const uint16_t* Wu;
};

template <int RT>
struct WsGeo {
    static constexpr int NS = ws_slots(RT);          // activation stage slots in LDS
    static constexpr int LA = NS - 1;                // stages of lead of the activation DMA
    static constexpr int PW = LA + WS_PWX;           // stages of lead of the weight loads
    static constexpr int NW = PW + 1;                // weight register sets
    static constexpr int NA = (RT + 1) / 2;          // LDS-DMA pieces per wave per stage (4 RT pieces of 1 KB over 8 waves)
    static constexpr int STAGE = RT * 2 * WS_SUB;    // hi sub-tiles then lo sub-tiles
    static constexpr int NSTEP = 2 * RT;             // (k-step, row tile) steps of 4 MFMAs
    static constexpr int NITEM = NA + 4;             // things a wave issues per stage: NA pieces, then 4 weight loads
    static constexpr int IPS = (NITEM + NSTEP - 1) / NSTEP;   // issued per step
    // loads younger than the last thing stage k needs, when r = nk - 1 - k stages remain after k.  A stage issues its
    // pieces A(s + LA) first, then its weight loads W(s + PW).  PW = LA + 1: W(k) is older than A(k), the wait is for A(k),
    // followed by W(k + 1) and the (LA - 1) later stages' issues; PW = LA: W(k) follows A(k) in the same stage, the wait is
    // for W(k), followed by the (LA - 1) later stages' issues only.
    static constexpr int younger(int r) {
        return WS_PWX ? 4 * (r < LA ? r : LA) + NA * (r < LA - 1 ? r : LA - 1) : (NA + 4) * (r < LA - 1 ? r : LA - 1);
    }
    static_assert(WS_PWX == 0 || WS_PWX == 1, "weight lead");
    static_assert(NW <= 4, "weight ring v[192:255] holds four sets");
    static_assert(NS >= 2 && NS <= WS_NSMAX && RT <= WS_RTMAX, "ring geometry");
    static_assert(younger(LA) < 64, "vmcnt range");
};

// One tile: prologue, K loop, epilogue.
template <int RT>
__device__ __forceinline__ void ws_tile(const VhGemmPsArgs& p, const WsTile& t, unsigned char* lds, const int lane,
                                        const int wid, const bool glu) {
    using G = WsGeo<RT>;
    constexpr bool NT = WS_NT != 0;
    constexpr int NS = G::NS, LA = G::LA, PW = G::PW, NW = G::NW, NA = G::NA;
    constexpr int FD = WS_FD < 2 ? 2 : WS_FD;
    const int nk = t.nk;

    // ---- this wave's weight rows ------------------------------------------------------------------------------
    // GLU: gate rows n0 + 16 wid + (l & 15) of W (fragment 0) and the same rows of W_up (fragment 1) -> output columns
    // n0 + 16 wid ..; plain: rows n0 + 32 wid + 16 f + (l & 15).
    const int lrow = lane & 15;
    int nr0 = glu ? t.n0 + 16 * wid + lrow : t.n0 + 32 * wid + lrow;
    int nr1 = glu ? nr0 : nr0 + 16;
    if (nr0 > p.N - 1) nr0 = p.N - 1;               // clamped rows: products never stored
    if (nr1 > p.N - 1) nr1 = p.N - 1;
    const uint32_t wcol = (uint32_t)(lane >> 4) * 16u;
    const uint32_t voff0 = (uint32_t)nr0 * (uint32_t)(p.ldw * 2) + wcol;
    const uint32_t voff1 = (uint32_t)nr1 * (uint32_t)(p.ldw * 2) + wcol;
    const unsigned char* const wb0 = ws_uniform(reinterpret_cast<const unsigned char*>(t.Wb) + (size_t)t.k0 * 128);
    const unsigned char* const wb1 = ws_uniform(reinterpret_cast<const unsigned char*>(glu ? t.Wu : t.Wb) + (size_t)t.k0 * 128);

    // ---- this wave's LDS-DMA pieces: plane wid & 1, 8-row half (wid >> 1) & 1 of row tiles (wid >> 2) + 2 j --------
    // (a wave whose last piece does not exist — odd RT — repeats its previous one: every wave issues NA per stage, so
    // the counted waits are the same for all)
    const int ap = wid & 1, ah8 = (wid >> 1) & 1, ai0 = wid >> 2;
    const int r16 = ah8 * 8 + (lane >> 3);
    const uint32_t acol = (uint32_t)(((lane & 7) ^ ((r16 >> 1) & 7)) * 16);
    uint32_t offa[NA];
    int adst[NA];                                    // byte offset of the piece inside a stage slot (wave-uniform)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int i = ai0 + 2 * j;
        if (i > RT - 1) i = (ai0 + 2 * (j - 1) >= 0 && j > 0) ? ai0 + 2 * (j - 1) : RT - 1;
        int m = t.m_begin + i * 16 + r16;
        if (m > t.m_end - 1) m = t.m_end - 1;
        const long src_row = p.a_rowidx ? p.a_rowidx[m] : m;
        offa[j] = (uint32_t)((size_t)src_row * p.lda * 2) + acol;
        adst[j] = (ap * RT + i) * WS_SUB + ah8 * 1024;
    }
    const unsigned char* const a_plane = ws_uniform(reinterpret_cast<const unsigned char*>(ap ? p.A_lo : p.A_hi) + (size_t)t.k0 * 128);
    // (v_readfirstlane -> SGPR read by a VMEM instruction needs 5 wait states the compiler does not add for an asm
    // statement: every base above is consumed after the address arithmetic below, far more than 5 instructions later;
    // the explicit nop keeps that true whatever the scheduler does)
    asm volatile("s_nop 4" ::: "memory");

    // fragment read offset inside a sub-tile for k-step ks: row r = lane & 15, chunk = ks * 4 + (lane >> 4)
    const int frag_base = (lrow >> 3) * 1024 + (lrow & 7) * 128;
    const int frag_x = (lrow >> 1) & 7;
    const int fo0 = frag_base + (((lane >> 4)) ^ frag_x) * 16;
    const int fo1 = frag_base + ((4 + (lane >> 4)) ^ frag_x) * 16;

    f32x4 acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    auto issue_w = [&](int kt, auto set_c, int q) __attribute__((always_inline)) {     // load q of stage kt into set SET
        constexpr int SET = decltype(set_c)::value;
        const size_t kb = (size_t)kt * 128;
        constexpr int R = WS_RING0 + SET * 16;       // set SET: [fragment 0 k-step 0, f0 k1, f1 k0, f1 k1]
        if (q == 0) ws_ldw<NT, R + 0, 0>(wb0 + kb, voff0);
        else if (q == 1) ws_ldw<NT, R + 4, 64>(wb0 + kb, voff0);
        else if (q == 2) ws_ldw<NT, R + 8, 0>(wb1 + kb, voff1);
        else ws_ldw<NT, R + 12, 64>(wb1 + kb, voff1);
    };
    auto issue_a = [&](int kt, int slot, int j) __attribute__((always_inline)) {
        if ((WS_ABLATE & 8) && ap) return;
        if (!(WS_ABLATE & 1)) ws_glds16(a_plane + (size_t)kt * 128, offa[j], lds + slot * G::STAGE + adst[j]);
    };

    // ---- prologue: [W(0);] then A(j), W(j + PWX) for j < LA — the order every later stage keeps -------------------------
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
    if (WS_PWX) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_w(0, S0{}, q);
    }
    if (0 < nk) {                                    // j = 0 (LA >= 1 always)
#pragma unroll
        for (int j = 0; j < NA; ++j) issue_a(0, 0, j);
    }
    if (WS_PWX < nk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_w(WS_PWX, std::integral_constant<int, WS_PWX>{}, q);
    }
    if (LA >= 2) {                                   // j = 1
        if (1 < nk) {
#pragma unroll
            for (int j = 0; j < NA; ++j) issue_a(1, 1, j);
        }
        if (1 + WS_PWX < nk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) issue_w(1 + WS_PWX, std::integral_constant<int, 1 + WS_PWX>{}, q);
        }
    }
    int slot_cur = 0;                                // slot of stage k
    int slot_fill = LA % NS;                         // slot of stage k + LA (= the slot stage k - 1 was read from)

    // ---- one stage ------------------------------------------------------------------------------------------------
    auto stage = [&](const int k, auto set_c) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
        constexpr int FSET = (SET + PW) % NW;        // the set stage k + PW goes to (= the set of stage k - 1)
        const int r = nk - 1 - k;
        // my pieces of stage k have landed (and with them, older in the queue, the weights of stage k)
        if (r >= LA) ws_wait<G::younger(LA)>();      // (younger() saturates at LA resp. LA - 1 remaining stages)
        else if (LA >= 2 && r == 1) ws_wait<G::younger(1)>();
        else ws_wait<0>();
        asm volatile("s_barrier" ::: "memory");      // stage k visible to every wave; stage k - 1 consumed by every wave
        __builtin_amdgcn_sched_barrier(0);
        const bool do_a = k + LA < nk, do_w = k + PW < nk;
        const unsigned char* ab = lds + slot_cur * G::STAGE;
        constexpr int R = WS_RING0 + SET * 16;
        const bf16x8_t w00 = ws_rdw<R + 0>(), w01 = ws_rdw<R + 4>();      // [fragment][k-step]
        const bf16x8_t w10 = ws_rdw<R + 8>(), w11 = ws_rdw<R + 12>();
        // activation fragments run FD - 1 steps (of 4 MFMAs = 64 matrix-pipe cycles) ahead of the MFMAs that consume them:
        // a ds_read_b128 issued one step ahead (r03 first form) arrives after ~150-300 cycles with eight waves reading
        bf16x8_t fh[FD], fl[FD];
#pragma unroll
        for (int s = 0; s < FD - 1 && s < G::NSTEP; ++s) {
            const int ks = s / RT, i = s % RT;
            fh[s] = *reinterpret_cast<const bf16x8_t*>(ab + i * WS_SUB + (ks ? fo1 : fo0));
            fl[s] = *reinterpret_cast<const bf16x8_t*>(ab + (RT + i) * WS_SUB + (ks ? fo1 : fo0));
        }
#pragma unroll
        for (int s = 0; s < G::NSTEP; ++s) {
            const int ks = s / RT, i = s % RT;
            if (s + FD - 1 < G::NSTEP) {
                const int ks1 = (s + FD - 1) / RT, i1 = (s + FD - 1) % RT;
                fh[(s + FD - 1) % FD] = *reinterpret_cast<const bf16x8_t*>(ab + i1 * WS_SUB + (ks1 ? fo1 : fo0));
                fl[(s + FD - 1) % FD] = *reinterpret_cast<const bf16x8_t*>(ab + (RT + i1) * WS_SUB + (ks1 ? fo1 : fo0));
            }
#pragma unroll
            for (int q = s * G::IPS; q < (s + 1) * G::IPS && q < G::NITEM; ++q) {      // this step's loads: pieces first
                if (q < NA) { if (do_a) issue_a(k + LA, slot_fill, q); }
                else if (do_w) issue_w(k + PW, std::integral_constant<int, FSET>{}, q - NA);
            }
            const bf16x8_t ah = fh[s % FD], al = fl[s % FD];
            if (!(WS_ABLATE & 4)) {
                const bf16x8_t wf0 = ks ? w01 : w00, wf1 = ks ? w11 : w10;
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0, ah, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1, ah, acc[i][1], 0, 0, 0);
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0, al, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1, al, acc[i][1], 0, 0, 0);
            } else {
                asm volatile("" ::"v"(ah), "v"(al), "v"(w00), "v"(w01), "v"(w10), "v"(w11));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        slot_cur = slot_cur + 1 == NS ? 0 : slot_cur + 1;
        slot_fill = slot_fill + 1 == NS ? 0 : slot_fill + 1;
    };

    for (int k = 0; k < nk; k += NW) {
        stage(k, S0{});
        if (k + 1 < nk) stage(k + 1, S1{});
        if (NW > 2) { if (k + 2 < nk) stage(k + 2, std::integral_constant<int, (NW > 2 ? 2 : 0)>{}); }
        if (NW > 3) { if (k + 3 < nk) stage(k + 3, std::integral_constant<int, (NW > 3 ? 3 : 0)>{}); }
    }
    asm volatile("s_barrier" ::: "memory");          // every wave is done with the LDS slots before the next tile refills them

    // ---- epilogue: acc[i][f][r] = out[token m_begin + 16 i + (lane & 15)][col0(f) + 4 (lane >> 4) + r] ----------------
    // (row tiles through a compile-time recursion: a `#pragma unroll` loop of this size was left rolled by hipcc once the
    // kernel held several instantiations, with the accumulators dumped to scratch and indexed dynamically)
    const int jrow = lane & 15, jc = (lane >> 4) * 4;
    auto row = [&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        const int m = t.m_begin + i * 16 + jrow;
        if (m >= t.m_end) return;
        const long orow = p.c_rowidx ? p.c_rowidx[m] : m;
        if (glu) {
            const int n = t.n0 + wid * 16 + jc;
            if (n >= p.N) return;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_f(acc[i][0][r]) * acc[i][1][r];
            const bool full = n + 3 < p.N;
            if (p.C) {
                float* cp = p.C + orow * p.ldc + n;
                if (full && ((p.ldc & 3) == 0)) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) cp[r] = v[r];
            }
            if (p.C_hi) {
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) split_bf16(v[r], hi[r], lo[r]);
                uint16_t* hp = p.C_hi + orow * p.ldc_split + n;
                uint16_t* lp = p.C_lo + orow * p.ldc_split + n;
                if (full && ((p.ldc_split & 3) == 0)) {
                    *reinterpret_cast<uint2*>(hp) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
                    *reinterpret_cast<uint2*>(lp) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
                } else {
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) { hp[r] = (uint16_t)hi[r]; lp[r] = (uint16_t)lo[r]; }
                }
            }
        } else {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int n = t.n0 + wid * 32 + f * 16 + jc;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float tv = acc[i][f][r];
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
                    uint32_t hi[4], lo[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) split_bf16(v[r], hi[r], lo[r]);
                    uint16_t* hp = p.C_hi + orow * p.ldc_split + n;
                    uint16_t* lp = p.C_lo + orow * p.ldc_split + n;
                    if (full && ((p.ldc_split & 3) == 0)) {
                        *reinterpret_cast<uint2*>(hp) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
                        *reinterpret_cast<uint2*>(lp) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
                    } else {
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) { hp[r] = (uint16_t)hi[r]; lp[r] = (uint16_t)lo[r]; }
                    }
                }
            }
        }
    };
    ws_static_for<0, RT>(row);
}

// amdgpu_num_vgpr(96): on gfx90a+ hipcc DOUBLES the requested number (unified VGPR + AGPR file), so 96 caps the
// compiler's own allocation at v0..v191; v[192:255] stay out of its reach (and the clobber below makes the kernel
// descriptor allocate them).
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) __attribute__((amdgpu_num_vgpr(WS_RING0 / 2)))
void k_gemm_ws(const VhGemmPsArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[WS_LDS];
    constexpr int RTMAX = WS_RTMAX;
    asm volatile("" ::: "v192", "v255");             // the weight ring: makes the kernel descriptor allocate all 256 VGPRs
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool glu = p.W_up != nullptr;
    const int NOUT = glu ? 128 : 256;

    const int E = p.group_off ? p.ngroups : 1;
    const int NT_ = (p.N + NOUT - 1) / NOUT;
    const int nk_total = p.K >> 6;
    auto rows_of = [&](int e) { return p.group_off ? (p.group_off[e + 1] - p.group_off[e]) : p.M; };
    auto mtiles_of = [&](int rows) { return (((rows + 15) >> 4) + RTMAX - 1) / RTMAX; };

    // ---- tile list and its partition (vh_gemm_ps.hip: experts by decreasing rows, one run of the list per XCD, the
    // partial last round cut along M, K split chosen here when the caller allows it) ------------------------------------
    int ord[8], n_exp = 0;
    for (int e = 0; e < E && e < 8; ++e) ord[n_exp++] = e;
    for (int i = 1; i < n_exp; ++i)
        for (int k = i; k > 0 && rows_of(ord[k]) > rows_of(ord[k - 1]); --k) { const int tmp = ord[k]; ord[k] = ord[k - 1]; ord[k - 1] = tmp; }
    int MT = 0;
    for (int i = 0; i < n_exp; ++i) MT += mtiles_of(rows_of(ord[i]));
    const int nb = gridDim.x >> 3;
    int KS = p.ksplit > 1 ? p.ksplit : 1;
    if (p.ksplit < 0) {
        int best = 1 << 30;
        for (int ks = 1; ks <= -p.ksplit && ks <= nk_total; ++ks) {
            const int Tx = (MT * NT_ * ks + 7) >> 3;
            const int Rr = Tx / nb, rr = Tx - Rr * nb;
            const int rounds16 = 16 * Rr + (rr == 0 ? 0 : (2 * rr <= nb ? 9 : 16));
            const int est = (rounds16 * 64) / ks + 4 * rounds16 + ks * ((48 * 64) / nk_total);   // as vh_gemm_ps.hip
            if (est < best) { best = est; KS = ks; }
        }
        if (p.nslab_out && blockIdx.x == 0 && threadIdx.x == 0) *p.nslab_out = KS;
    }
    const int T = MT * NT_ * KS;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int g0 = (int)(((long)T * xcd) >> 3), g1 = (int)(((long)T * (xcd + 1)) >> 3);
    const int Tx = g1 - g0;
    const int R = Tx / nb, r = Tx - R * nb;
    const bool split_tail = r > 0 && 2 * r <= nb;
    for (int it = 0; it <= R; ++it) {
        int g, half = -1;
        if (it < R) g = g0 + it * nb + j;
        else if (r == 0) break;
        else if (split_tail) { if (j >= 2 * r) break; g = g0 + R * nb + (j >> 1); half = j & 1; }
        else { if (j >= r) break; g = g0 + R * nb + j; }
        int e = 0, li = g, rows = 0, mt = 0, oi = 0;
        for (; oi < n_exp; ++oi) {
            e = ord[oi];
            rows = rows_of(e);
            mt = mtiles_of(rows);
            const int cnt = mt * NT_ * KS;
            if (li < cnt) break;
            li -= cnt;
        }
        if (oi == n_exp) break;
        const int mi = li % mt;
        li /= mt;
        const int nt = li % NT_, ks = li / NT_;
        const int nrt = (rows + 15) >> 4;
        const int rtper = (nrt + mt - 1) / mt;
        const int off_e = p.group_off ? p.group_off[e] : 0;
        WsTile t;
        t.m_begin = off_e + mi * rtper * 16;
        t.m_end = min(off_e + rows, t.m_begin + rtper * 16);
        if (t.m_begin >= t.m_end) continue;
        t.rt = (t.m_end - t.m_begin + 15) >> 4;
        if (half >= 0) {
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
        t.Wu = glu ? p.W_up + (size_t)e * p.w_group_stride : nullptr;
        switch (t.rt) {
#define WS_CASE(RT) case RT: ws_tile<RT>(p, t, lds, lane, wid, glu); break;
#ifdef WS_DEV_RTS      // development builds: only the row-tile counts of the micro-benchmark (compile time)
            WS_DEV_RTS
#else
            WS_CASE(1) WS_CASE(2) WS_CASE(3) WS_CASE(4) WS_CASE(5) WS_CASE(6) WS_CASE(7) WS_CASE(8) WS_CASE(9)
            WS_CASE(10) WS_CASE(11) WS_CASE(12) WS_CASE(13) WS_CASE(14)
#endif
#undef WS_CASE
            default: break;
        }
    }
}

}  // namespace

int vhk_gemm_ws(hipStream_t st, const VhGemmPsArgs& a, int grid, bool nt) {
    (void)nt;
    hipLaunchKernelGGL(k_gemm_ws, dim3(grid), dim3(512), 0, st, a);
    return 0;
}
