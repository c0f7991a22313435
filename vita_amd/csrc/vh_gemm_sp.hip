// vh_gemm_sp.hip — the weight-streaming GEMM on pre-split activations with SPECIALISED waves (r04; ps_cfg = 2).
// Same job, LDS image, tile list, persistent schedule and epilogue as vh_gemm_ps.hip (HF MixtralExperts,
// modeling_mixtral.py:57-93 as reached from web_demo/vllm_tools/vllm_file/mixtral.py:405-422) — what changes is WHO does what.
//
// Why.  In vh_gemm_ps.hip all 8 waves of a block load AND multiply, one block barrier per K stage.  Its counters (r02 / r03:
// profiles/r03_pmc_moe_sq.txt) say the waves are parked 31 % of their cycles on s_waitcnt / the barrier and the matrix pipe is
// busy 46 %: the weight-staging waves carry 16 global loads + 16 ds_write_b128 + the wait for the HBM data on top of MORE
// row tiles than the other half (ceil(rt/2) vs floor), and every stall of any wave is a stall of all at the next barrier;
// ablations put the loads alone at 510 us and the matrix work alone at 334 us against 642 us for both: the phases add.
//
// Here a block is 12 waves, 3 per SIMD (168 VGPRs each):
//   waves 0-7   MFMA waves (2 M x 4 N as before): per stage ONE barrier, fragment reads, MFMAs.  No vector-memory
//               instruction, no LDS store, no vmcnt wait anywhere in their loop;
//   waves 8, 9  weight stagers by stage parity: wave 8+g owns the stages s = 1 + g (mod 2) (stage 0: half each): 32 KB of
//               weights global -> 128 staging VGPRs right behind barrier s-3 (as soon as the registers are free), registers ->
//               LDS behind barrier s-1: two stage times for the HBM latency, and it issues nothing in between, so the compiler's
//               vmcnt(0) in front of the stores is exact; its stall for HBM happens while the MFMA waves work;
//   waves 10,11 activation DMA (global_load_lds, inline asm as in vh_gemm_ps.hip): the 8-row half 0 / 1 of every row tile,
//               both planes, one stage ahead (L2 hits).
// Tried on top of this and removed again (r04, profiles/r04_sp_pingpong_ab.txt): two barriers per stage with the MFMA waves of a
// SIMD half a stage apart, so that one of them always has prefetched fragments at a barrier ("ping-pong"): correct, 534-538 us
// against 520-532 — the stage time is set by what the CU can keep in flight, not by the fragment prologue.
// Each SIMD therefore holds two MFMA waves (one of either M half: the matrix pipe sees all row tiles of the stage) plus one
// loader whose issue slots interleave with theirs.  Numerics are those of vh_gemm_ps.hip (same fragments, same order of the
// MFMAs per accumulator).
#include "vh_common.h"
#include "vh_kernels.h"

namespace {

#ifndef PS_SCHED
#define PS_SCHED 1         // tile -> block: cost-ordered grid stride (vh_gemm_ps_inl.h): 631 -> 613 us skewed, engine prefill 35.8 -> 35.0 ms
#endif
#include "vh_gemm_ps_inl.h"

#define SP_WAVES 12
#define SP_NSLOT 4
#ifndef SP_PRIO
#define SP_PRIO 1          // s_setprio level of the MFMA waves (static).  0 / 1 / 3 measured (profiles/r04_sp_prio.txt): 520 / 505 / 502 us gate|up
#endif

template <bool GLU, int RTMAX, int RTW, int RTA, bool NTW>
__device__ __forceinline__ void run_tile_sp(const VhGemmPsArgs& p, const TileCtx& t, unsigned char* lds, const int lane,
                                            const int wid) {
    constexpr int A_BUF = RTMAX * 2 * MG_SUB;    // one activation stage: hi sub-tiles then lo sub-tiles
    constexpr int W_BASE = 2 * A_BUF;
    constexpr int LDS_BYTES = 2 * A_BUF + SP_NSLOT * MG_SLOT;
    constexpr int NPA = RTW + RTA;               // row tiles of the tile
    const int lrow = lane >> 3;

    if (wid < 8) {
        // ================================ MFMA waves ===========================================================
        const int wm = wid >> 2, wn = wid & 3;
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

        auto stage = [&](const int k, auto rte_c) __attribute__((always_inline)) {
            constexpr int RTE = decltype(rte_c)::value;                  // row tiles this wave multiplies
            constexpr int NSTEP = 2 * (RTE > 0 ? RTE : 1);               // (k-step, row tile) steps of 8 MFMAs per stage
            const unsigned char* ab = lds + (k & 1) * A_BUF;
            const int slot_w = ((2 * k) & 3) + (wn >> 1);                // stage k: sub-tiles 0-7 in slot (2k)%4, 8-15 in the next
            const unsigned char* wb = lds + W_BASE + slot_w * MG_SLOT + (wn & 1) * 4 * MG_SUB;
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
            // pin the order: fragment reads one step ahead of the MFMAs that consume them (hipcc otherwise sinks every read
            // to just before its MFMAs and waits lgkmcnt(0) in front of each group of four)
            __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);               // 8 weight + 2 activation fragment reads
            StepOrder<0, NSTEP, 0, 0>::pin();
        };
        auto loop = [&](auto rte_c) __attribute__((always_inline)) {
            for (int k = 0; k < t.nk; ++k) {
                __builtin_amdgcn_s_barrier();         // stage k visible to every wave; stage k-1 fully consumed
                __builtin_amdgcn_sched_barrier(0);
                stage(k, rte_c);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#if SP_PRIO
        __builtin_amdgcn_s_setprio(SP_PRIO);          // static: the MFMA waves win issue arbitration against their SIMD's loader wave
#endif
        // the M-half branch is OUTSIDE the K loop (matching barrier counts; a branch inside made hipcc duplicate the accumulators)
        if (wm == 0) loop(std::integral_constant<int, RTW>{});
        else loop(std::integral_constant<int, RTA>{});
        __builtin_amdgcn_s_barrier();                 // every wave is done with the rings
        tile_epilogue<GLU, RTMAX, RTW, SP_WAVES, LDS_BYTES, true>(p, t, lds, lane, wid, wm, wn, acc);
        return;
    }

    if (wid < 10) {
        // ================================ weight stagers (stage parity g) =======================================
        const int g = wid - 8;
        const unsigned char* w_gate = reinterpret_cast<const unsigned char*>(t.Wb);
        const unsigned char* w_up = reinterpret_cast<const unsigned char*>(GLU ? t.Wu : t.Wb);
        const uint32_t ldw2 = (uint32_t)(p.ldw * 2);
        // piece q = 2 * s16 + u: sub-tile s16 (16 weight rows), 8-row half u; lane -> row u*8 + lrow, 16-byte chunk (lane & 7)
        // permuted on the source side (chunk ^ (row >> 1)) as in vh_gemm_ps.hip
        uint32_t wcol[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) wcol[u] = (uint32_t)(((lane & 7) ^ (((u * 8 + lrow) >> 1) & 7)) * 16);
        u32x4 wreg[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) wreg[q] = u32x4{0u, 0u, 0u, 0u};
        auto w_load = [&](int kt) __attribute__((always_inline)) {       // stage kt -> registers
            const int kc = kt < t.nk ? kt : t.nk - 1;                    // clamped re-load at the tile's end: never stored
            const size_t kb = (size_t)(t.k0 + kc) * 128;
            // the 32 per-lane offsets are recomputed per stage ON PURPOSE (a few VALU each in a wave that has nothing else to
            // issue): hoisted out of the K loop as invariants they would sit in 32 VGPRs next to the 128 staging registers
            int lr = lrow;
            asm volatile("" : "+v"(lr));
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int s16 = q >> 1, u = q & 1;
                const unsigned char* base = ((GLU && (s16 & 2)) ? w_up : w_gate) + kb;
                int n = GLU ? t.n0 + (s16 >> 2) * 32 + (s16 & 1) * 16 + u * 8 + lr : t.n0 + s16 * 16 + u * 8 + lr;
                if (n > p.N - 1) n = p.N - 1;                            // clamped rows: products never stored
                const uint32_t o = (uint32_t)n * ldw2 + wcol[u];
                if (NTW) wreg[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + o));
                else wreg[q] = *reinterpret_cast<const u32x4*>(base + o);
            }
        };
        unsigned char* const wr_dst0 = lds + W_BASE + lane * 16;
        auto w_store = [&](int kt) __attribute__((always_inline)) {      // registers -> the LDS image of stage kt
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int s16 = q >> 1, u = q & 1;
                *reinterpret_cast<u32x4*>(wr_dst0 + (((2 * kt) & 3) + (s16 >> 3)) * MG_SLOT + (s16 & 7) * MG_SUB + u * 1024) = wreg[q];
            }
        };
        // Stage 0 is staged by BOTH stagers (half each: it is needed at once); after that stager g owns the stages 1 + g, 3 + g, ...:
        // registers <- stage s right after barrier s-3 (the barrier behind its previous store), registers -> LDS right after barrier
        // s-1.  With the ownership shifted like this both stagers run the SAME loop (stager 1 enters it one barrier later), so the
        // 128 staging registers have one definition and one use per iteration and no control-flow merge (a first version with one
        // loop per parity made hipcc spill the whole staging set at the loop entries).
        {
            const size_t kb = (size_t)t.k0 * 128;
            int lr = lrow;
            asm volatile("" : "+v"(lr));
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {
                const int s16 = 8 * g + (qq >> 1), u = qq & 1;             // sub-tiles 8g .. 8g+7 = slot g of stage 0
                const unsigned char* base = ((GLU && (s16 & 2)) ? w_up : w_gate) + kb;
                int n = GLU ? t.n0 + (s16 >> 2) * 32 + (s16 & 1) * 16 + u * 8 + lr : t.n0 + s16 * 16 + u * 8 + lr;
                if (n > p.N - 1) n = p.N - 1;
                const uint32_t o = (uint32_t)n * ldw2 + wcol[u];
                if (NTW) wreg[qq] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + o));
                else wreg[qq] = *reinterpret_cast<const u32x4*>(base + o);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {
                const int s16 = qq >> 1, u = qq & 1;                      // position inside slot g
                *reinterpret_cast<u32x4*>(wr_dst0 + g * MG_SLOT + s16 * MG_SUB + u * 1024) = wreg[qq];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        w_load(1 + g);
        __builtin_amdgcn_sched_barrier(0);
        int k = g;
        if (g == 1) {                                  // stager 1 owns nothing at barrier 0
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // The reload is issued one barrier AFTER the store, so the loads have one stage time to land and each stager has loads in
        // flight only every other stage.  Reloading right behind the store (two stage times of lead, 64 KB of weights in flight per
        // CU at all times) was measured SLOWER — 549-555 / 631 us (uniform / skewed) against 503-531 / 594: the CU's in-flight
        // request budget (~56 KB) is shared with the activation DMA (profiles/r04_sp_lead_sched_ab.txt).
        for (; k < t.nk; k += 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // barrier k: stage k-1 is consumed, its slots take stage k+1
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < t.nk) {
                w_store(k + 1);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();          // barrier k+1
                __builtin_amdgcn_sched_barrier(0);
                w_load(k + 3);                         // (clamped past the tile's end: never stored)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // clamped tail loads
        __builtin_amdgcn_s_barrier();
    } else {
        // ================================ activation DMA (8-row half u of every row tile, both planes) =============
        const int u = wid - 10;
        const int r16 = u * 8 + lrow;
        const int c8 = (lane & 7) ^ ((r16 >> 1) & 7);
        uint32_t offa[RTMAX];
#pragma unroll
        for (int i = 0; i < RTMAX; ++i) {
            int m = t.m_begin + i * 16 + r16;
            if (m > t.m_end - 1) m = t.m_end - 1;
            const long src_row = p.a_rowidx ? p.a_rowidx[m] : m;
            offa[i] = (uint32_t)(((size_t)src_row * p.lda + c8 * 8) * 2);
        }
        const unsigned char* a_hi = reinterpret_cast<const unsigned char*>(p.A_hi);
        const unsigned char* a_lo = reinterpret_cast<const unsigned char*>(p.A_lo);
        unsigned char* const a_dst0 = lds + u * 1024;
        auto a_stage = [&](int kt) __attribute__((always_inline)) {
            const int kc = kt < t.nk ? kt : t.nk - 1;
            const size_t kb = (size_t)(t.k0 + kc) * 128;
            unsigned char* dst = a_dst0 + (kt & 1) * A_BUF;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                glds16<false>(a_hi + kb, offa[i], dst + i * MG_SUB);
                glds16<false>(a_lo + kb, offa[i], dst + (RTMAX + i) * MG_SUB);
            }
        };
        a_stage(0);
        for (int k = 0; k < t.nk; ++k) {
            wait_vm<0>();                             // stage k has landed
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            a_stage(k + 1);                           // (clamped at the tile's end: lands in the buffer nobody reads again)
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
    }
    f32x4 none[RTW][4];
    tile_epilogue<GLU, RTMAX, RTW, SP_WAVES, LDS_BYTES, false>(p, t, lds, lane, wid, 0, 0, none);
}


template <bool GLU, int RTMAX, bool NTW>
__global__ __launch_bounds__(64 * SP_WAVES) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_gemm_sp(const VhGemmPsArgs p) {
    constexpr int A_BUF = RTMAX * 2 * MG_SUB;
    static_assert(2 * A_BUF + SP_NSLOT * MG_SLOT <= MG_LDS && (RTMAX % 2) == 0, "LDS budget / ring geometry");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * A_BUF + SP_NSLOT * MG_SLOT];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for_each_tile<GLU, RTMAX>(p, [&](const TileCtx& t) __attribute__((always_inline)) {
        switch (t.rt) {
#define SP_CASE(RT)                                                                                        \
    case RT:                                                                                               \
        if constexpr (RTMAX >= RT) run_tile_sp<GLU, RTMAX, (RT + 1) / 2, RT / 2, NTW>(p, t, lds, lane, wid); \
        break;
            SP_CASE(1) SP_CASE(2) SP_CASE(3) SP_CASE(4) SP_CASE(5) SP_CASE(6)
            SP_CASE(7) SP_CASE(8) SP_CASE(9) SP_CASE(10) SP_CASE(11) SP_CASE(12)
#undef SP_CASE
            default: break;
        }
    });
}

}  // namespace

// arguments already checked by vhk_gemm_ps (vh_gemm_ps.hip)
int vhk_gemm_sp(hipStream_t st, const VhGemmPsArgs& a, int grid, bool nt) {
    if (a.W_up) {
        if (nt) hipLaunchKernelGGL((k_gemm_sp<true, 12, true>), dim3(grid), dim3(64 * SP_WAVES), 0, st, a);
        else hipLaunchKernelGGL((k_gemm_sp<true, 12, false>), dim3(grid), dim3(64 * SP_WAVES), 0, st, a);
    } else {
        if (nt) hipLaunchKernelGGL((k_gemm_sp<false, 12, true>), dim3(grid), dim3(64 * SP_WAVES), 0, st, a);
        else hipLaunchKernelGGL((k_gemm_sp<false, 12, false>), dim3(grid), dim3(64 * SP_WAVES), 0, st, a);
    }
    return 0;
}
