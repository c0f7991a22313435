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

// round-to-nearest-even f32 -> bf16 bit pattern (inputs are finite here).
__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

// x ~= hi + lo with hi = bf16(x), lo = bf16(x - hi); |x - hi - lo| <= 2^-17 |x|.
__device__ __forceinline__ void split_bf16(float x, uint32_t& hi, uint32_t& lo) {
    hi = f32_to_bf16_rne(x);
    float r = x - __uint_as_float(hi << 16);
    lo = f32_to_bf16_rne(r);
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

// ---- wave / block reductions ----------------------------------------------
// Inside a row of 16 lanes the butterfly runs on DPP modifiers (quad_perm xor 1, xor 2, row_half_mirror, row_mirror):
// VALU-rate, no LDS crossbar.  (__shfl_xor compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): ~100 cycles of exposed
// latency per step — 28 of them per 32-key tile made the attention softmax longer than its MFMAs.)  Only the two
// cross-row steps of a full-wave reduction still go through ds_bpermute.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
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

// Sum NV values per thread over a 256-thread block (4 waves). `red` is LDS
// scratch of >= 4*NV floats. Result broadcast to every thread. Two barriers.
template <int NV>
__device__ __forceinline__ void block256_sum(float (&v)[NV], float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = wave_sum(v[i]);
        if (lane == 0) red[wid * NV + i] = s;
    }
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
