// vh_common.h — shared device helpers for the gfx950 (CDNA4, wave64) kernels.
//
// Numeric contract of the whole library ("exact mode"):
//   * weights are stored in HBM as bf16 (the checkpoint's stored dtype,
//     reference web_demo/vllm_tools/model_weight_file/config.json:40),
//   * every activation, the KV cache and every accumulator is fp32,
//   * a GEMM with more than one token row feeds the bf16 MFMA with the
//     activation split into hi + lo bf16 halves (x = hi + lo to 2^-17), so
//     results track an fp32 reference on the same bf16-rounded weights to
//     ~1e-6 relative — this is what lets the parity tests use the north-star
//     tolerance (logits within 1e-3, greedy ids bit-exact) against a pure
//     fp32 oracle instead of a bf16-emulating one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VH_WAVE 64

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA 16x16x32 bf16 operand (4 VGPR)

// ---- bf16 <-> f32 ---------------------------------------------------------
__device__ __forceinline__ float bf16_lo_to_f32(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi_to_f32(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// round-to-nearest-even f32 -> bf16 bit pattern: gfx950's converter (v_cvt_pk_bf16_f32).  (r01-r02 did this in integer
// arithmetic: ~5 VALU per conversion, 14 per hi/lo split — which made the general GEMM's operand staging, 16 values per thread
// and K-tile, cost more issue cycles than its 32 MFMAs.)
typedef __attribute__((ext_vector_type(2))) float vh_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 vh_bf16x2;
__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    const __bf16 b = (__bf16)f;
    return (uint32_t)__builtin_bit_cast(unsigned short, b);
}

// x ~= hi + lo with hi = bf16(x), lo = bf16(x - hi); |x - hi - lo| <= 2^-17 |x|.
__device__ __forceinline__ void split_bf16(float x, uint32_t& hi, uint32_t& lo) {
    hi = f32_to_bf16_rne(x);
    float r = x - __uint_as_float(hi << 16);
    lo = f32_to_bf16_rne(r);
}
// The same for two values at once, results PACKED (a in the low half): 5 VALU per pair (cvt_pk, and, lshl, pk_add, cvt_pk).
__device__ __forceinline__ void split_bf16_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    const vh_f32x2 v = {a, b};
    const vh_bf16x2 h = __builtin_convertvector(v, vh_bf16x2);
    const vh_f32x2 r = v - __builtin_convertvector(h, vh_f32x2);
    const vh_bf16x2 l = __builtin_convertvector(r, vh_bf16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

// 8 bf16 (one 16-byte chunk) dotted with 8 fp32 activations.
__device__ __forceinline__ float dot8_bf16_f32(const uint4& w, const float* x) {
    float a = 0.f;
    a = fmaf(bf16_lo_to_f32(w.x), x[0], a);
    a = fmaf(bf16_hi_to_f32(w.x), x[1], a);
    a = fmaf(bf16_lo_to_f32(w.y), x[2], a);
    a = fmaf(bf16_hi_to_f32(w.y), x[3], a);
    a = fmaf(bf16_lo_to_f32(w.z), x[4], a);
    a = fmaf(bf16_hi_to_f32(w.z), x[5], a);
    a = fmaf(bf16_lo_to_f32(w.w), x[6], a);
    a = fmaf(bf16_hi_to_f32(w.w), x[7], a);
    return a;
}

// Streamed-once weight load: non-temporal so the stream does not evict the
// L2-resident activations (MI355X_MICROARCH "nt-weights": -5..10 % per layer).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ uint4 ld_weight16(const void* p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

#ifndef VH_BLOCKSUM_TRANSPOSE
#define VH_BLOCKSUM_TRANSPOSE 1
#endif
// ---- wave / block reductions ----------------------------------------------
// Inside a row of 16 lanes the butterfly runs on DPP modifiers (quad_perm xor 1, xor 2, row_half_mirror, row_mirror):
// VALU-rate, no LDS crossbar.  (__shfl_xor compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): ~100 cycles of exposed
// latency per step — 28 of them per 32-key tile made the attention softmax longer than its MFMAs.)  Only the two
// cross-row steps of a full-wave reduction still go through ds_bpermute.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    // full row / bank masks + bound_ctrl: no lane of these controls reads out of its row, and the `old` operand is dead, so
    // the move folds into the consuming VALU op (v_max_f32_dpp ...); with bound_ctrl off hipcc kept "v_mov 0; v_mov_dpp; v_max"
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
#define VH_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define VH_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define VH_DPP_HALF_MIRROR 0x141  // lane i <-> 7 - i inside each half row
#define VH_DPP_MIRROR 0x140       // lane i <-> 15 - i inside the row
// reduce inside aligned groups of 16 lanes (MFMA 16x16 D-layout rows); every lane gets the result.
__device__ __forceinline__ float grp16_sum(float v) {
    v += dpp_mov<VH_DPP_XOR1>(v);
    v += dpp_mov<VH_DPP_XOR2>(v);
    v += dpp_mov<VH_DPP_HALF_MIRROR>(v);
    v += dpp_mov<VH_DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float grp16_max(float v) {
    v = fmaxf(v, dpp_mov<VH_DPP_XOR1>(v));
    v = fmaxf(v, dpp_mov<VH_DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<VH_DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<VH_DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = grp16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = grp16_max(v);
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// ---- many values per thread: transposing wave reduction ------------------------------------------------------------
// wave_sum costs 6 exchange steps PER VALUE.  With NV values per lane the butterfly can halve the value set at every
// step instead: partners exchange the half they do not keep and add, so after the six steps lane L holds the wave total
// of ONE value (index returned by multi_idx) — sum_k ceil(NV / 2^k) ~ NV exchanges in all instead of 6 NV.  Pairings:
// row_mirror, row_half_mirror, quad_perm xor 2 / xor 1 (DPP, VALU rate), then lane ^ 16, lane ^ 32 (ds_bpermute).
// The summation tree differs from wave_sum's, so kernels pick one or the other for all their values.
template <int CTRL>
__device__ __forceinline__ float multi_xchg_dpp(float send) { return dpp_mov<CTRL>(send); }
__host__ __device__ constexpr int multi_bit(int step) { return step == 0 ? 3 : step == 1 ? 2 : step == 2 ? 1 : step == 3 ? 0 : step; }
template <int N, int STEP>
struct MultiReduce {
    static constexpr int H = (N + 1) / 2;
    __device__ static __forceinline__ float run(float (&v)[N], int lane) {
        // Partners must hold the SAME value subset, i.e. agree in every lane bit consumed by the earlier steps, and differ
        // in this step's bit: row_mirror (i <-> 15-i, decided by bit 3) first — everyone still holds everything —, then
        // row_half_mirror (i <-> 7-i: same bit 3, decided by bit 2), quad xor 2 (bit 1), quad xor 1 (bit 0), lane ^ 16, ^ 32.
        constexpr int BIT = multi_bit(STEP);
        const bool upper = (lane >> BIT) & 1;
        float nv[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const float lo = v[i];
            const float hi = (i + H < N) ? v[i + H] : 0.f;
            const float keep = upper ? hi : lo;
            const float send = upper ? lo : hi;
            float recv;
            if (STEP == 0) recv = multi_xchg_dpp<VH_DPP_MIRROR>(send);
            else if (STEP == 1) recv = multi_xchg_dpp<VH_DPP_HALF_MIRROR>(send);
            else if (STEP == 2) recv = multi_xchg_dpp<VH_DPP_XOR2>(send);
            else if (STEP == 3) recv = multi_xchg_dpp<VH_DPP_XOR1>(send);
            else recv = __shfl_xor(send, 1 << STEP, 64);
            nv[i] = keep + recv;
        }
        if constexpr (STEP == 5) return nv[0];
        else return MultiReduce<H, STEP + 1>::run(nv, lane);
    }
};
// index of the value whose wave total lane `lane` holds after MultiReduce<NV, 0>::run (>= NV: padding, ignore)
template <int NV>
__device__ __forceinline__ int multi_idx(int lane) {
    int idx = 0, n = NV, real = NV;             // n: (padded) array length at this step, real: entries that are not padding
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int h = (n + 1) / 2;
        if ((lane >> multi_bit(k)) & 1) { idx += h; real = real > h ? real - h : 0; }
        else real = real < h ? real : h;
        n = h;
    }
    return real > 0 ? idx : NV;                 // NV = "holds padding"
}
// Block (256 threads = 4 waves) totals of NV <= 64 values per thread, left in LDS: tot[i] = sum over the block of v[i].
// `red` needs 4 * NV floats, `tot` NV floats.  Every thread may read tot[] after the call.  Three barriers.
template <int NV>
__device__ __forceinline__ void block256_multi_sum(float (&v)[NV], float* red, float* tot) {
    static_assert(NV >= 1 && NV <= 64, "one value per lane at the end");
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float w = MultiReduce<NV, 0>::run(v, lane);
    const int idx = multi_idx<NV>(lane);
    if (idx < NV) red[wid * NV + idx] = w;
    __syncthreads();
    if ((int)threadIdx.x < NV) tot[threadIdx.x] = red[threadIdx.x] + red[NV + threadIdx.x] + red[2 * NV + threadIdx.x] + red[3 * NV + threadIdx.x];
    __syncthreads();
}

// Sum NV values per thread over a 256-thread block (4 waves). `red` is LDS
// scratch of >= 4*NV floats. Result broadcast to every thread. Two barriers.
template <int NV>
__device__ __forceinline__ void block256_sum(float (&v)[NV], float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#if VH_BLOCKSUM_TRANSPOSE
    // wave level by the transposing butterfly (~NV exchanges instead of 6 NV), block level through LDS as before
    const float w = MultiReduce<NV, 0>::run(v, lane);
    const int idx = multi_idx<NV>(lane);
    if (idx < NV) red[wid * NV + idx] = w;
#else
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = wave_sum(v[i]);
        if (lane == 0) red[wid * NV + i] = s;
    }
#endif
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = red[i] + red[NV + i] + red[2 * NV + i] + red[3 * NV + i];
    __syncthreads();
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// activation ids shared by the GEMM / norm epilogues (vita_hip.h VH_ACT_*)
#define VH_ACT_NONE 0
#define VH_ACT_GELU 1
#define VH_ACT_RELU 2
#define VH_ACT_SILU 3
__device__ __forceinline__ float apply_act(float x, int act) {
    if (act == VH_ACT_GELU) return gelu_erf(x);
    if (act == VH_ACT_RELU) return fmaxf(x, 0.f);
    if (act == VH_ACT_SILU) return silu_f(x);
    return x;
}

// number of compute units of the current device (256 on MI355X)
inline int vh_num_cus() {
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
