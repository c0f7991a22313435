// vh_decode.hip — batch-1 decode kernels for the Mixtral backbone.
//
// One generated token touches every active weight exactly once, so decode is a pure
// HBM stream (25.66 GB/token at TP=1).  Five kernels per layer, each a row-parallel
// GEMV whose prologue recomputes the (tiny, L2-resident) activation-side work instead
// of paying a kernel boundary for it:
//
//   k_dec_gemv<NORM>   x = x_in + delta ; RMSNorm ; fused Q|K|V GEMV           (K20,K21)
//   k_dec_attn         RoPE(q,k_new) ; KV-cache append ; split-KV GQA attention ;
//                      last-arriver merge of the split partials                 (K22,K23)
//   k_dec_gemv<!NORM>  O-projection GEMV -> delta_attn                          (K24)
//   k_dec_gateup       x = x_in + delta ; RMSNorm ; router softmax/top-2 (on device, no
//                      host sync) ; gate|up GEMV of the two chosen experts ; SiLU*up (K25,K26)
//   k_dec_down         down GEMV of the two experts, routing-weighted sum -> delta_moe (K26)
//   k_dec_lmhead       final RMSNorm ; LM-head GEMV ; per-block argmax          (K27)
//   k_dec_select       global argmax ; append token ; pos++ ; next embedding    (K28,K18)
//
// GEMV shape: a 256-thread block owns R output rows; thread t owns 16-byte chunks
// c = t + 256*j of every row (a wave reads 1 KiB contiguous per load instruction),
// keeps its slice of the fp32 activation vector in registers and puts all R*NJ weight
// loads in flight before anything else (deep vmcnt, no LDS round trip — guide §5 "GEMV /
// M<=16 decode weights"); the persistent kernels (gate|up, LM head) loop over row groups and rely on
// the co-resident blocks of a CU to overlap one block's reduction with another's loads.  Output rows are reduced wave->LDS->thread, no
// atomics, so results are deterministic.  Residual adds are deferred into the consumer's
// prologue ("delta" buffers) so that under tensor parallelism the same kernels run
// unchanged with an all-reduce on the delta buffer between them.
//
// r06: the attention block of a layer (fused QKV -> attention -> O projection) is ONE launch, k_dec_ablk: its rows and tiles are
// work items of 2 persistent blocks per CU, taken in a fixed order (item i on block i mod grid); an item that needs another
// item's output polls it as tagged 8-byte {tag, fp32} granules (VhGranVec, agent-scope atomics) with its own weights / K-V tile
// already in flight.  The per-thread arithmetic and the reduction order are those of the three separate kernels (k_dec_gemv<NORM>,
// k_dec_attn, k_dec_gemv<!NORM>, kept as the per-operator entries and as vh_tune("dec_fused", 0)): the two forms are bit-identical.
//
// Reference semantics restated: transformers/models/mixtral/modeling_mixtral.py
// (MixtralRMSNorm, MixtralAttention + apply_rotary_pos_emb, MixtralTopKRouter,
// MixtralExperts) as called from vita/model/language_model/vita_mixtral.py:158-173.
#include "vh_common.h"
#include "vh_kernels.h"

namespace {

// ---- tensor-parallel exchange fused into the kernels (VhXchg, vh_kernels.h) ---------------------------------------------
typedef unsigned long long xu64;
#define VH_XCHG_SPIN_LIMIT (1u << 26)
// producer: element n of my partial vector -> slot `rank` of every rank's receive region (the data is the flag)
__device__ __forceinline__ void xchg_put(const VhXchg& px, int n, float v) {
    const xu64 g = ((xu64)px.tag << 32) | (xu64)__float_as_uint(v);
    if (px.loopback) {       // one rank plays them all: my value into slot `rank`, the peers' (zero) contributions into theirs
        const xu64 z = (xu64)px.tag << 32;
        for (int p = 0; p < px.world; ++p)
            __hip_atomic_store(reinterpret_cast<xu64*>(px.local + (size_t)p * px.cap + n), p == px.rank ? g : z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    for (int p = 0; p < px.world; ++p)
        __hip_atomic_store(reinterpret_cast<xu64*>(px.peer[p] + (size_t)px.rank * px.cap + n), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// consumer, step 1 (BEFORE the block's own weight loads, so that nothing of its own is queued in front of the polls): the
// first nred blocks sum their slice over the ranks, in rank order, and publish it.
// r06: ONE element per thread and all `world` slots of it polled TOGETHER (r03-r05: a pair per thread and the 2 x world granules one
// after the other — sixteen dependent round trips to the uncached receive buffer per exchange: the consumer kernels ran 7 us longer
// than without an exchange, profiles/r06 call 1); a look that finds a tag missing re-reads all slots after a short sleep.
__device__ __forceinline__ void xchg_reduce(const VhXchg& cx) {
    if (cx.world == 0 || (int)blockIdx.x >= cx.nred) return;
    const int per = (cx.count + cx.nred - 1) / cx.nred;
    const int n1 = min(cx.count, ((int)blockIdx.x + 1) * per);
    for (int n = blockIdx.x * per + threadIdx.x; n < n1; n += blockDim.x) {
        xu64 g[8];
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r < cx.world) g[r] = __hip_atomic_load(reinterpret_cast<const xu64*>(cx.local + (size_t)r * cx.cap + n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r < cx.world) ok = ok && ((unsigned)(g[r] >> 32) == cx.tag);
            if (ok) break;
            if ((spins & 1023u) == 0 && __hip_atomic_load(cx.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (++spins > VH_XCHG_SPIN_LIMIT) { __hip_atomic_store(cx.err, 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < cx.world) s += __uint_as_float((unsigned)g[r]);          // rank order: the same sum on every rank
        // published as a tagged granule in the GEMV layout: the readers' sweep (gran_read_gemv) needs neither a counter nor a drain
        __hip_atomic_store(reinterpret_cast<xu64*>(cx.reduced_g) + vhk_gran_pos_gemv(n), ((xu64)cx.tag << 32) | (xu64)__float_as_uint(s),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// consumer, step 2 (AFTER the block's weight loads are in flight): load_add_norm reads the summed vector with gran_read_gemv.
// Residency: the nred reducer blocks are the FIRST blocks of the consumer grid (lowest indices: dispatched first, and they wait
// only for the peers' pushes), so a reading block never keeps a reducer from being scheduled when each rank owns its GPU; ranks
// SHARING a device (tests) can starve each other — a waiting consumer grid holds the CUs a peer's producer needs —, which is why
// they take one all-reduce kernel per exchange instead, and why every spin is bounded and ends in the sticky error word.

// ---- granule vectors between the work items of the fused attention-block launch (VhGranVec, vh_kernels.h) ---------------------
// Every wait below is WAVE-collective (all 64 lanes run the same number of polls; exits are decided with __all) and bounded: a
// producer that never publishes ends in the engine's error word (code 7), not in a hang.
__device__ __forceinline__ size_t gran_pos_gemv(int n) { return vhk_gran_pos_gemv(n); }
__device__ __forceinline__ void gran_put(const VhGranVec& gv, size_t pos, float v) {
    __hip_atomic_store(reinterpret_cast<xu64*>(gv.g) + pos, ((xu64)gv.tag << 32) | (xu64)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ xu64 gran_ld(const VhGranVec& gv, size_t pos) {
    return __hip_atomic_load(reinterpret_cast<const xu64*>(gv.g) + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one more unsuccessful poll: true = give up (wave-uniform: spins is, and the error word is one address)
__device__ __forceinline__ bool gran_spin_fail(unsigned& spins, int* err) {
    if ((spins & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;   // someone else failed
    if (++spins > VH_GRAN_SPIN_LIMIT) {
        __hip_atomic_store(err, 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    __builtin_amdgcn_s_sleep(8);
    return false;
}
// N granules per lane (positions idx[]), e.g. the two halves of a head's q row for the rotate-half RoPE
template <int N>
__device__ __forceinline__ void gran_getn(const VhGranVec& gv, const int (&idx)[N], float (&out)[N]) {
    unsigned spins = 0;
    for (;;) {
        xu64 x[N];
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = gran_ld(gv, (size_t)idx[i]);
#pragma unroll
        for (int i = 0; i < N; ++i) { ok = ok && ((unsigned)(x[i] >> 32) == gv.tag); out[i] = __uint_as_float((unsigned)x[i]); }
        if (__all(ok)) return;
        if (gran_spin_fail(spins, gv.err)) return;
    }
}
// a GEMV block's slice of a layout-1 vector: thread t gets elements [8 (t + 256 j), + 8), j < NJ (zeros past K).  Called by ALL
// 256 threads of the block (block barrier inside).
// Polling discipline (r05, first form measured 174 against 209 tok/s): a consumer kernel is resident for microseconds before its
// producer publishes, and 2048 + 1536 waves re-reading ONE granule every 0.3 us queue up behind each other on that line's L2
// channel — in front of the producer's own store to it.  So: (1) ONE wave per block polls, the other three wait at the barrier;
// (2) it polls ONE granule per look (lane-uniform address: a single 8-byte request), a DIFFERENT element for every block
// (spread over the channels), about every microsecond; (3) only then does every wave sweep its own granules.
template <int NJ>
__device__ __forceinline__ void gran_read_gemv(const VhGranVec& gv, int K, float (&v)[NJ][8]) {
    const int t = threadIdx.x;
    unsigned spins = 0;
    if (t < 64) {
        const size_t sent = gran_pos_gemv((int)((blockIdx.x * 1031u + 17u) % (unsigned)K));
        for (;;) {
            if ((unsigned)(gran_ld(gv, sent) >> 32) == gv.tag) break;
            if (gran_spin_fail(spins, gv.err)) break;
            __builtin_amdgcn_s_sleep(8);              // + the 8 of gran_spin_fail: ~0.5 us between looks (<= ~1000 poller waves in all; s_sleep 2 + 1 measured no faster: r06 call 9)
        }
    }
    __syncthreads();
    spins = 0;
    // this thread's granules, re-read until every tag of the wave matches (the producers publish within ~1 us of each other).
    // Addresses are a block-uniform base (scalar registers) + one per-lane offset: with per-lane 64-bit pointers the sixteen
    // addresses alone took 32 vector registers and the capped instantiations spilled.
    const int nj_in = (K + 2047) >> 11;                   // chunk slots j that hold data for at least one thread
    const int tt = ((t + (nj_in - 1) * 256) * 8 < K) ? t : 0;   // threads past K in the LAST slot re-read lane 0's granule (discarded)
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j < nj_in) {                                  // block-uniform
                const xu64* base = reinterpret_cast<const xu64*>(gv.g) + (size_t)(j * 8) * 256;
                const int tj = (j == nj_in - 1) ? tt : t;
                xu64 x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = __hip_atomic_load(base + e * 256 + tj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool in = (t + j * 256) * 8 < K;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ok = ok && ((unsigned)(x[e] >> 32) == gv.tag);
                    v[j][e] = in ? __uint_as_float((unsigned)x[e]) : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
            }
        }
        if (__all(ok)) return;
        if (gran_spin_fail(spins, gv.err)) return;
    }
}

// ---- activation slice handling ------------------------------------------------------
template <int NJ>
__device__ __forceinline__ void load_x(const float* __restrict__ x, int K, float (&xr)[NJ][8]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c * 8 < K) {
            const float4 a = reinterpret_cast<const float4*>(x)[c * 2];
            const float4 b = reinterpret_cast<const float4*>(x)[c * 2 + 1];
            xr[j][0] = a.x; xr[j][1] = a.y; xr[j][2] = a.z; xr[j][3] = a.w;
            xr[j][4] = b.x; xr[j][5] = b.y; xr[j][6] = b.z; xr[j][7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) xr[j][i] = 0.f;
        }
    }
}

// x = x_in + delta (optionally stored to x_out by block 0), then x <- x * w_norm; returns this thread's
// share of sum(x^2).  The scalar rsqrt(mean(x^2)+eps) of MixtralRMSNorm (modeling_mixtral.py:143-148) commutes
// with the GEMV, so callers apply it to the reduced dot products: one block reduction instead of two.  The three activation-side vectors are loaded TOGETHER (one L2 round trip, not three
// dependent ones: the separate load_x / add_x / scale_by_norm_weight sequence cost ~1 us per kernel).
template <int NJ>
__device__ __forceinline__ float load_add_norm(const float* __restrict__ x_in, const float* __restrict__ delta,
                                               const float* __restrict__ norm_w, float* __restrict__ x_out, int K,
                                               float (&xr)[NJ][8], const VhXchg* cx = nullptr, const bool store_block = false) {
    // cx (world > 0): `delta` is the result of a fused exchange — read as granules (block barrier inside: every thread calls this)
    // x_out (nullable) is stored by block 0 of the launch, or — store_block — by whichever block the caller passes a non-null x_out to
    float dg[NJ][8];
    const bool fused = cx != nullptr && cx->world != 0;          // block-uniform
    if (fused) gran_read_gemv<NJ>(VhGranVec{cx->reduced_g, cx->tag, cx->err}, K, dg);
    f32x4 xa[NJ][2], da[NJ][2], na[NJ][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = threadIdx.x + j * 256;
        const int cc = (c * 8 < K) ? c : 0;   // clamped: unconditional loads, masked below
        xa[j][0] = reinterpret_cast<const f32x4*>(x_in)[cc * 2];
        xa[j][1] = reinterpret_cast<const f32x4*>(x_in)[cc * 2 + 1];
        na[j][0] = reinterpret_cast<const f32x4*>(norm_w)[cc * 2];
        na[j][1] = reinterpret_cast<const f32x4*>(norm_w)[cc * 2 + 1];
        if (fused) {
            da[j][0] = f32x4{dg[j][0], dg[j][1], dg[j][2], dg[j][3]};
            da[j][1] = f32x4{dg[j][4], dg[j][5], dg[j][6], dg[j][7]};
        } else if (delta) {
            da[j][0] = reinterpret_cast<const f32x4*>(delta)[cc * 2];
            da[j][1] = reinterpret_cast<const f32x4*>(delta)[cc * 2 + 1];
        } else {
            da[j][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            da[j][1] = da[j][0];
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = threadIdx.x + j * 256;
        const bool ok = c * 8 < K;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x4 v = xa[j][hh] + da[j][hh];
            if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (x_out && (store_block || blockIdx.x == 0) && ok) reinterpret_cast<f32x4*>(x_out)[c * 2 + hh] = v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ss = fmaf(v[i], v[i], ss);
                xr[j][hh * 4 + i] = v[i] * na[j][hh][i];
            }
        }
    }
    return ss;
}

// R rows of W (bf16) against the register-resident x, in two phases so the weight loads are in
// flight BEFORE the (L2-resident) activation loads and prologue math.  rows[r] must be valid
// pointers (callers clamp out-of-range rows and drop the result).
template <int NJ, int R>
__device__ __forceinline__ void gemv_issue(const uint16_t* const (&rows)[R], int K, uint4 (&w)[R][NJ]) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = threadIdx.x + j * 256;
            if (c * 8 < K) w[r][j] = ld_weight16(rows[r] + (size_t)c * 8);
            else w[r][j] = make_uint4(0, 0, 0, 0);
        }
}
template <int NJ, int R>
__device__ __forceinline__ void gemv_fma(const uint4 (&w)[R][NJ], const float (&xr)[NJ][8], float (&acc)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) a += dot8_bf16_f32(w[r][j], xr[j]);
        acc[r] = a;
    }
}

// ---- K_A / K_C: (residual add + RMSNorm +) row-parallel GEMV -------------------------------
// NORM=true : out = rsqrt(mean(x^2)+eps) * W (x*w_norm), x = x_in + delta      (fused QKV)
// NORM=false: out = W x_in                                                    (O projection)
// (the per-operator form; inside a decode step the same arithmetic runs as items of k_dec_ablk below)
template <int NJ, int R, bool NORM>
__global__ __launch_bounds__(256) void k_dec_gemv(const float* __restrict__ x_in, const float* __restrict__ delta,
                                                  float* __restrict__ x_out, const float* __restrict__ norm_w,
                                                  float eps, const uint16_t* __restrict__ W, int N, int K,
                                                  float* __restrict__ out, const VhXchg xc) {
    // xc: NORM = true: consumer of a fused exchange (delta = xc.reduced); NORM = false: producer (outputs are pushed)
    __shared__ float red[4 * (R + 1)];
    if (NORM) xchg_reduce(xc);
    const int n0 = blockIdx.x * R;
    const uint16_t* rows[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rows[r] = W + (size_t)min(n0 + r, N - 1) * K;
    uint4 w[R][NJ];
    gemv_issue<NJ, R>(rows, K, w);

    float xr[NJ][8];
    float vals[R + 1];
    if (NORM) {
        vals[R] = load_add_norm<NJ>(x_in, delta, norm_w, x_out, K, xr, &xc);
    } else {
        load_x<NJ>(x_in, K, xr);
        vals[R] = 0.f;
    }
    float acc[R];
    gemv_fma<NJ, R>(w, xr, acc);
#pragma unroll
    for (int r = 0; r < R; ++r) vals[r] = acc[r];
    block256_sum<R + 1>(vals, red);
    const float inv = NORM ? rsqrtf(vals[R] / (float)K + eps) : 1.0f;
    if (threadIdx.x < R && n0 + threadIdx.x < N) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) if (threadIdx.x == r) v = vals[r];
        if (!NORM && xc.world) xchg_put(xc, n0 + threadIdx.x, v * inv);
        else out[n0 + threadIdx.x] = v * inv;
    }
}

// ---- K_B: RoPE + KV append + split-KV GQA attention + in-kernel merge -----------------------
// grid (nkv, nsplit): block (h, sp) owns keys [64 sp, 64 sp + 64) of KV head h — exactly one LDS
// tile, shared by the G = nq/nkv query heads of that KV head (wave w < G serves head h*G + w;
// lanes index keys for QK^T and head dims, 2 per lane, for PV).  The host sizes nsplit from its
// mirror of the position, so no block is empty.  Each block publishes (m, l, o) partials; the
// LAST block to arrive for a KV head merges them and writes the attention output, so the
// O-projection reads 16 KB instead of re-merging the partials in each of its blocks.
// Hand-off of the partials (r06): every partial word is written with an 8-byte agent-scope atomic store (write-through) and
// read by the merging block with 8-byte agent-scope atomic loads (L1-bypassing) — guide G16 "8-B agent atomics both sides":
// every storing wave drains (vmcnt(0)), one relaxed ticket per block, NO release / acquire fence (r01-r05 paid one of each,
// ~1.7 us apiece on the critical path of a kernel that is pure latency).  head_dim is 128 (config.json:16-44).
#define DA_KT 64
#define DA_KSTR 132
struct DecAttnTile { f32x4 k[8], v[8]; };      // one 64-key K / V tile in flight: 8 x 16 B of each per thread
// `table` (nullable): paged KV cache — logical 64-key block sp of this sequence lives in physical page table[sp] of the
// pool (one page = one tile of this kernel); null = the contiguous single-sequence layout (page sp).
// Step 1 of an attention block: put the K/V tile loads in flight (8 x 16 B each per thread).  Unconditional loads from
// clamped rows into NATIVE vector registers: with the loads under a branch and the HIP float4 struct
// as the staging type, hipcc parked the K registers in scratch memory and waited after every K/V pair
// (8 dependent round trips instead of 16 loads in flight).  Rows that do not exist are zeroed in step 3.
__device__ __forceinline__ void dec_attn_issue(const int h, const int sp, const float* __restrict__ kcache,
                                               const float* __restrict__ vcache, const int* __restrict__ table, int max_ctx,
                                               DecAttnTile& t) {
    const int p0 = (table ? table[sp] : sp) * DA_KT;                 // first physical row of this tile's page
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int row = idx >> 5, c4 = idx & 31;
        const int key = min(p0 + row, max_ctx - 1);
        t.k[i] = reinterpret_cast<const f32x4*>(kcache + ((size_t)h * max_ctx + key) * 128)[c4];
        t.v[i] = reinterpret_cast<const f32x4*>(vcache + ((size_t)h * max_ctx + key) * 128)[c4];
    }
}
__device__ __forceinline__ void part_store2(float* p, float a, float b) {
    __hip_atomic_store(reinterpret_cast<xu64*>(p), ((xu64)__float_as_uint(b) << 32) | (xu64)__float_as_uint(a), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 part_load2(const float* p) {
    const xu64 x = __hip_atomic_load(reinterpret_cast<const xu64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)x), __uint_as_float((unsigned)(x >> 32)));
}
// Steps 2-5 of one attention block (h, sp of nsplit) whose tile loads were issued by dec_attn_issue.  Returns true in the block
// that merged the partials of its KV head and wrote attn_out rows [h*G*128, (h+1)*G*128) (block-uniform).
// GR: qkv arrives as LINEAR granules (gq) from fused-QKV items of the same launch that may still be running; the merged
// output leaves as GEMV-layout granules (gout) for the O-projection items.
template <bool GR>
__device__ __forceinline__ bool dec_attn_finish(const int h, const int sp, const int nsplit, const DecAttnTile& tile,
                                                const float* __restrict__ qkv, float* __restrict__ kcache,
                                                float* __restrict__ vcache, const int pos, const int* __restrict__ table,
                                                const float* __restrict__ rope_cos,
                                                const float* __restrict__ rope_sin, float* __restrict__ part_o,
                                                float* __restrict__ part_ml, int* __restrict__ cnt,
                                                float* __restrict__ attn_out, int nq, int nkv, int max_ctx,
                                                int max_splits, float scale, const VhGranVec gq, const VhGranVec gout) {
    __shared__ __attribute__((aligned(16))) float q_s[4][128];
    __shared__ __attribute__((aligned(16))) float kn_s[128];
    __shared__ __attribute__((aligned(16))) float vn_s[128];
    __shared__ __attribute__((aligned(16))) float Kt[DA_KT * DA_KSTR];     // the K tile, then (r06) the V tile: 36 KB per block instead of 69,
    float* const Vt = Kt;                                                  // so four blocks of the fused attention-block launch fit a CU
    __shared__ int last_s;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int G = nq / nkv;
    // the position comes from the host mirror (it sized this grid): no dependent load before the K/V tiles
    const int k0 = sp * DA_KT;
    const int k1 = min(pos + 1, k0 + DA_KT);
    const int head = h * G + wid;
    const int p0 = (table ? table[sp] : sp) * DA_KT;                 // first physical row of this tile's page

    // 2. rotate-half RoPE (modeling_mixtral.py:203-241): out = x*cos + rotate_half(x)*sin
    const float c = rope_cos[(size_t)pos * 64 + lane], s = rope_sin[(size_t)pos * 64 + lane];
    if (wid < G) {
        float ab[2];
        if (GR) {
            const int idx[2] = {head * 128 + lane, head * 128 + 64 + lane};
            gran_getn<2>(gq, idx, ab);
        } else {
            ab[0] = qkv[head * 128 + lane]; ab[1] = qkv[head * 128 + 64 + lane];
        }
        const float a = ab[0], b = ab[1];
        q_s[wid][lane] = a * c - b * s;
        q_s[wid][lane + 64] = b * c + a * s;
    }
    const bool has_new = (pos >= k0) && (pos < k1);
    if (has_new && wid == 3) {  // wave 3 is idle for G<4 and cheap otherwise
        float kv4[4];
        if (GR) {
            const int kb0 = nq * 128 + h * 128, vb0 = (nq + nkv) * 128 + h * 128;
            const int idx[4] = {kb0 + lane, kb0 + 64 + lane, vb0 + lane, vb0 + 64 + lane};
            gran_getn<4>(gq, idx, kv4);
        } else {
            const float* kp = qkv + (size_t)nq * 128 + h * 128;
            const float* vp = qkv + (size_t)(nq + nkv) * 128 + h * 128;
            kv4[0] = kp[lane]; kv4[1] = kp[lane + 64]; kv4[2] = vp[lane]; kv4[3] = vp[lane + 64];
        }
        const float a = kv4[0], b = kv4[1];
        const float ka = a * c - b * s, kb = b * c + a * s;
        const float va = kv4[2], vb = kv4[3];
        kn_s[lane] = ka; kn_s[lane + 64] = kb;
        vn_s[lane] = va; vn_s[lane + 64] = vb;
        float* kc = kcache + ((size_t)h * max_ctx + p0 + (pos - k0)) * 128;   // has_new: pos lies in this tile
        float* vc = vcache + ((size_t)h * max_ctx + p0 + (pos - k0)) * 128;
        kc[lane] = ka; kc[lane + 64] = kb;
        vc[lane] = va; vc[lane + 64] = vb;
    }
    __syncthreads();

    // 3. K tile -> LDS (the new token's row comes from LDS, not from the cache; rows past the context are 0); the V tile waits in
    // registers until the scores are done with the buffer
    f32x4 vtile[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + i * 256;
        const int row = idx >> 5, c4 = idx & 31;
        const int key = k0 + row;
        f32x4 kv = tile.k[i], vv = tile.v[i];
        if (key == pos) {
            kv = reinterpret_cast<const f32x4*>(kn_s)[c4];
            vv = reinterpret_cast<const f32x4*>(vn_s)[c4];
        } else if (key >= k1) {
            kv = f32x4{0.f, 0.f, 0.f, 0.f};
            vv = kv;
        }
        *reinterpret_cast<f32x4*>(&Kt[row * DA_KSTR + c4 * 4]) = kv;
        vtile[i] = vv;
    }
    __syncthreads();

    // 4. scores and softmax statistics for this tile; then the V tile takes the K tile's place; then PV
    float p = 0.f, m = 0.f, l = 0.f;
    if (wid < G) {
        const bool valid = (k0 + lane) < k1;
        float sc = 0.f;
        const float4* qr = reinterpret_cast<const float4*>(q_s[wid]);
        const float4* kr = reinterpret_cast<const float4*>(&Kt[lane * DA_KSTR]);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float4 qq = qr[i], kk = kr[i];
            sc = fmaf(qq.x, kk.x, sc);
            sc = fmaf(qq.y, kk.y, sc);
            sc = fmaf(qq.z, kk.z, sc);
            sc = fmaf(qq.w, kk.w, sc);
        }
        sc = valid ? sc * scale : -INFINITY;
        m = wave_max(sc);                                // finite: the host never launches an empty tile
        p = valid ? __expf(sc - m) : 0.f;
        l = wave_sum(p);
    }
    __syncthreads();                                     // every wave is done with the K tile
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + i * 256;
        *reinterpret_cast<f32x4*>(&Vt[(idx >> 5) * 128 + (idx & 31) * 4]) = vtile[i];
    }
    __syncthreads();
    if (wid < G) {
        float2 o = make_float2(0.f, 0.f);
#pragma unroll
        for (int kk = 0; kk < DA_KT; ++kk) {
            const float pk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), kk));
            const float2 vv = reinterpret_cast<const float2*>(&Vt[kk * 128])[lane];
            o.x = fmaf(pk, vv.x, o.x);
            o.y = fmaf(pk, vv.y, o.y);
        }
        part_store2(part_o + ((size_t)head * max_splits + sp) * 128 + 2 * lane, o.x, o.y);
        if (lane == 0) part_store2(part_ml + ((size_t)head * max_splits + sp) * 2, m, l);
    }

    // 5. publish; the last arriver of this KV head merges (placement-independent hand-off: write-through stores, every wave
    // drained, one relaxed ticket; the merger reads with L1-bypassing loads)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(&cnt[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = (ticket == nsplit - 1) ? 1 : 0;
        if (ticket == nsplit - 1) __hip_atomic_store(&cnt[h], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
    if (!last_s) return false;
    if (wid < G) {
        // Merge of the nsplit partials of this head.  Every partial word is an L1-bypassing (sc1) load, ~0.7 us each when waited for
        // one at a time — r06's first form of this loop did exactly that (a loop-carried max, then four loads per iteration) and a
        // 9-split merge cost more than the tile's scores and PV together.  Now: lane s loads (m, l) of split s — ONE round trip for the
        // statistics —, the output partials come eight loads at a time, and the sums run in the OLD order (ascending split,
        // one fma chain) with the lane's values broadcast by readlane: same bits as r01-r05.
        const float* ml = part_ml + (size_t)head * max_splits * 2;
        const float* po = part_o + (size_t)head * max_splits * 128;
        float M = -INFINITY;
        for (int s0 = 0; s0 < nsplit; s0 += 64) {
            const int sl = s0 + lane;
            const float mv = sl < nsplit ? part_load2(ml + sl * 2).x : -INFINITY;
            M = fmaxf(M, wave_max(mv));
        }
        float den = 0.f;
        float2 num = make_float2(0.f, 0.f);
        for (int s0 = 0; s0 < nsplit; s0 += 64) {
            const int sl = s0 + lane;
            const float2 mlv = sl < nsplit ? part_load2(ml + sl * 2) : make_float2(-INFINITY, 0.f);
            const float wl = sl < nsplit ? __expf(mlv.x - M) : 0.f;
            const int cnt = min(64, nsplit - s0);
            for (int k0 = 0; k0 < cnt; k0 += 8) {
                float2 ov[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) ov[u] = part_load2(po + (size_t)(s0 + min(k0 + u, cnt - 1)) * 128 + 2 * lane);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (k0 + u < cnt) {                  // (wave-uniform)
                        const float wgt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl), k0 + u));
                        const float lv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mlv.y), k0 + u));
                        den = fmaf(wgt, lv, den);
                        num.x = fmaf(wgt, ov[u].x, num.x);
                        num.y = fmaf(wgt, ov[u].y, num.y);
                    }
                }
            }
        }
        const float inv = 1.0f / den;
        if (GR) {
            gran_put(gout, gran_pos_gemv(head * 128 + 2 * lane), num.x * inv);
            gran_put(gout, gran_pos_gemv(head * 128 + 2 * lane + 1), num.y * inv);
        } else {
            reinterpret_cast<float2*>(attn_out + (size_t)head * 128)[lane] = make_float2(num.x * inv, num.y * inv);
        }
    }
    return true;
}

__global__ __launch_bounds__(256) void k_dec_attn(const float* __restrict__ qkv, float* __restrict__ kcache,
                                                  float* __restrict__ vcache, const int pos, const int* __restrict__ table,
                                                  const float* __restrict__ rope_cos,
                                                  const float* __restrict__ rope_sin, float* __restrict__ part_o,
                                                  float* __restrict__ part_ml, int* __restrict__ cnt,
                                                  float* __restrict__ attn_out, int nq, int nkv, int max_ctx,
                                                  int max_splits, float scale) {
    DecAttnTile tile;
    dec_attn_issue(blockIdx.x, blockIdx.y, kcache, vcache, table, max_ctx, tile);
    dec_attn_finish<false>(blockIdx.x, blockIdx.y, gridDim.y, tile, qkv, kcache, vcache, pos, table, rope_cos, rope_sin, part_o,
                           part_ml, cnt, attn_out, nq, nkv, max_ctx, max_splits, scale, VhGranVec{}, VhGranVec{});
}

// ---- the attention block of a layer as ONE launch: fused QKV GEMV -> split-KV attention -> O projection ----------------------------
// A batch-1 decode layer used to be five dependent launches; QKV, attention and the O projection are short enough that the head
// and tail of each launch — dispatch, first-byte latency of the weight / K-V loads, prologue, drain, the boundary — cost as
// much as their bytes (TP = 1: 28.5 us for 84 MB that the stream moves in 14; one rank of TP = 8: 18 us for 10.5 MB).  Here the three
// are the blocks of ONE launch, by block index:
//   [0, nQ)               RQ rows of the fused q|k|v matrix each           -> q|k|v as LINEAR granules (gq)
//   [nQ, nQ + nA)         one 64-key tile of one KV head each (h, split)   -> partials; the last arriver of a head merges -> GEMV-layout granules (ga)
//   [nQ + nA, grid)       8 rows of the O projection each                  -> out (stored, or pushed to the peers: px)
// A block waits only for blocks with a LOWER index (attention for the QKV rows, O for the merged attention output), and the
// hardware dispatches a launch's blocks in index order: whatever else occupies the chip — another rank or another engine
// process on the same device included — the waited-for blocks were dispatched before the waiting one and depend on nothing
// behind them, so every wait ends (it is bounded anyway and ends in the error word).  A first form with persistent multi-role
// blocks (2 per CU, attention on the highest block ids) deadlocked when four ranks shared one device: every rank's resident
// O-phase blocks waited for attention blocks that found no free slot (r06 calls 1-2).  What the fusion buys: ONE launch ramp and
// boundary instead of three; a block's weights / K-V tile are requested the moment it starts, BEFORE it waits for its input
// granules: O blocks are dispatched as QKV blocks retire and have their weights in flight or landed when the attention publishes.
// Arithmetic: exactly k_dec_gemv<NORM> / k_dec_attn / k_dec_gemv<!NORM> (same per-thread accumulation, same reduction tree).
template <int NJ, int R>
__device__ __forceinline__ void ablk_issue_rows(const uint16_t* __restrict__ W, const int n0, const int N, const int K, uint4 (&w)[R][NJ]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint16_t* row = W + (size_t)min(n0 + r, N - 1) * K;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = threadIdx.x + j * 256;
            w[r][j] = (c * 8 < K) ? ld_weight16(row + (size_t)c * 8) : make_uint4(0, 0, 0, 0);
        }
    }
}
template <int NJ, int NJO, int RQ>
__global__ __launch_bounds__(256, 4) void k_dec_ablk(const VhDecAblk a) {
    constexpr int RO = 8;
    constexpr int RM = RQ > RO ? RQ : RO;
    __shared__ float red[4 * (RM + 1)];
    const int KO = a.nq * 128;
    const int nQ = (a.nqkv + RQ - 1) / RQ, nA = a.nkv * a.nsplit;
    const int b = blockIdx.x;
    if (b < nQ) {
        // ---- fused-QKV rows (the first nred of these blocks also reduce a fused exchange) ----
        xchg_reduce(a.cx);
        uint4 w[RQ][NJ];
        ablk_issue_rows<NJ, RQ>(a.Wqkv, b * RQ, a.nqkv, a.H, w);
        float xr[NJ][8];
        float vals[RQ + 1];
        vals[RQ] = load_add_norm<NJ>(a.x_in, a.delta, a.norm_w, a.x_out, a.H, xr, &a.cx);
        float acc[RQ];
        gemv_fma<NJ, RQ>(w, xr, acc);
#pragma unroll
        for (int r = 0; r < RQ; ++r) vals[r] = acc[r];
        block256_sum<RQ + 1>(vals, red);
        const float inv = rsqrtf(vals[RQ] / (float)a.H + a.eps);
        const int n0 = b * RQ;
        if (threadIdx.x < RQ && n0 + threadIdx.x < a.nqkv) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < RQ; ++r) if (threadIdx.x == r) v = vals[r];
            gran_put(a.gq, (size_t)(n0 + threadIdx.x), v * inv);
        }
    } else if (b < nQ + nA) {
        // ---- one attention tile ----
        const int ai = b - nQ;
        DecAttnTile tile;
        dec_attn_issue(ai / a.nsplit, ai % a.nsplit, a.kcache, a.vcache, a.table, a.max_ctx, tile);
        dec_attn_finish<true>(ai / a.nsplit, ai % a.nsplit, a.nsplit, tile, nullptr, a.kcache, a.vcache, a.pos, a.table, a.rope_cos,
                              a.rope_sin, a.part_o, a.part_ml, a.cnt, nullptr, a.nq, a.nkv, a.max_ctx, a.max_splits, a.scale, a.gq, a.ga);
    } else {
        // ---- O-projection rows ----
        const int n0 = (b - nQ - nA) * RO;
        uint4 w[RO][NJO];
        ablk_issue_rows<NJO, RO>(a.Wo, n0, a.H, KO, w);
        float xo[NJO][8];
        gran_read_gemv<NJO>(a.ga, KO, xo);      // waits for the merged attention output (this block's weights are in flight)
        float vals[RO];
        gemv_fma<NJO, RO>(w, xo, vals);
        block256_sum<RO>(vals, red);
        if (threadIdx.x < RO && n0 + threadIdx.x < a.H) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < RO; ++r) if (threadIdx.x == r) v = vals[r];
            if (a.px.world) xchg_put(a.px, n0 + threadIdx.x, v);
            else a.out[n0 + threadIdx.x] = v;
        }
    }
}

// ---- K_D: residual add + RMSNorm + router + gate|up GEMV + SiLU*up ------------------
// Router (modeling_mixtral.py:96-111): logits = x_n @ Wg^T ; softmax fp32 ; top-2 ;
// renormalise.  Every block recomputes it (8 rows, L2-resident) so the expert ids never
// leave the device.  route_out = {e0, e1, bits(w0), bits(w1)}.  Blocks are persistent: each
// loops over 2*RP-row groups (RP gate + RP up rows of one expert) and relies on the co-resident
// blocks of its CU for the overlap of one block's reduction with another's loads (a second register
// buffer per block measured slower, 91 vs 84 us: it costs two waves per SIMD of occupancy; r01, git history).
template <int NJ, int RP>
__device__ __forceinline__ void dec_gateup_body(const int bi, const int gn, const float* __restrict__ x_in,
                                                const float* __restrict__ delta, float* __restrict__ x_out,
                                                const float* __restrict__ norm_w, float eps, const uint16_t* __restrict__ Wg, int E,
                                                const uint16_t* __restrict__ W1, const uint16_t* __restrict__ W3, int I, int K,
                                                int* __restrict__ route_out, float* __restrict__ hbuf, const VhXchg& cx, float* red) {
    float xr[NJ][8];
    float inv;
    int e0 = 0, e1 = 0;
    {
        const uint16_t* rrows[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) rrows[e] = Wg + (size_t)min(e, E - 1) * K;
        uint4 wr[8][NJ];
        gemv_issue<NJ, 8>(rrows, K, wr);
        float vals[9];
        vals[8] = load_add_norm<NJ>(x_in, delta, norm_w, bi == 0 ? x_out : nullptr, K, xr, &cx, /*store_block=*/true);
        float lg[8];
        gemv_fma<NJ, 8>(wr, xr, lg);
#pragma unroll
        for (int e = 0; e < 8; ++e) vals[e] = lg[e];
        block256_sum<9>(vals, red);
        inv = rsqrtf(vals[8] / (float)K + eps);
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) { vals[e] *= inv; if (e < E) mx = fmaxf(mx, vals[e]); }
        float pr[8], sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { pr[e] = (e < E) ? expf(vals[e] - mx) : 0.f; sum += pr[e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) pr[e] = pr[e] / sum;
        float b0 = -1.f, b1 = -1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < E && pr[e] > b0) { b0 = pr[e]; e0 = e; }
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < E && e != e0 && pr[e] > b1) { b1 = pr[e]; e1 = e; }
        if (bi == 0 && threadIdx.x == 0) {
            const float t = b0 + b1;
            route_out[0] = e0; route_out[1] = e1;
            route_out[2] = __float_as_int(b0 / t); route_out[3] = __float_as_int(b1 / t);
        }
    }

    const int per_slot = I / RP;
    const int n_iter = 2 * per_slot;
    auto rows_of = [&](int it, const uint16_t* (&rows)[2 * RP]) {
        const int slot = it / per_slot;
        const int i0 = (it - slot * per_slot) * RP;
        const int e = slot ? e1 : e0;
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            rows[r] = W1 + ((size_t)e * I + i0 + r) * K;
            rows[RP + r] = W3 + ((size_t)e * I + i0 + r) * K;
        }
    };
    const uint16_t* rows[2 * RP];
    // (r06: issuing the NEXT group's loads right behind this group's FMAs — they would land under the block reduction — was measured
    // and lost: the kernel grew from 118 to 162 registers (hipcc keeps both weight sets live) and 78.5 -> 82.7 us at TP = 1,
    // 19.2 -> 20.3 at one rank's TP = 8 shard; the CU's other resident blocks already provide that overlap)
    for (int it = bi; it < n_iter; it += gn) {
        uint4 w[2 * RP][NJ];
        rows_of(it, rows);
        gemv_issue<NJ, 2 * RP>(rows, K, w);
        float acc[2 * RP];
        gemv_fma<NJ, 2 * RP>(w, xr, acc);
        block256_sum<2 * RP>(acc, red);
        if (threadIdx.x < RP) {
            float g = 0.f, u = 0.f;
#pragma unroll
            for (int r = 0; r < RP; ++r) if (threadIdx.x == r) { g = acc[r]; u = acc[RP + r]; }
            const int slot = it / per_slot;
            const int i0 = (it - slot * per_slot) * RP;
            hbuf[(size_t)slot * I + i0 + threadIdx.x] = silu_f(g * inv) * (u * inv);
        }
    }
}
template <int NJ, int RP>
__global__ __launch_bounds__(256) void k_dec_gateup(const float* __restrict__ x_in, const float* __restrict__ delta,
                                                    float* __restrict__ x_out, const float* __restrict__ norm_w,
                                                    float eps, const uint16_t* __restrict__ Wg, int E,
                                                    const uint16_t* __restrict__ W1, const uint16_t* __restrict__ W3,
                                                    int I, int K, int* __restrict__ route_out,
                                                    float* __restrict__ hbuf, const VhXchg cx) {
    __shared__ float red[4 * 9];
    xchg_reduce(cx);
    dec_gateup_body<NJ, RP>(blockIdx.x, gridDim.x, x_in, delta, x_out, norm_w, eps, Wg, E, W1, W3, I, K, route_out, hbuf, cx, red);
}

// ---- K_E: down GEMV of both experts, routing-weighted sum ---------------------------
template <int NJ, int R>
__global__ __launch_bounds__(256) void k_dec_down(const float* __restrict__ hbuf, const int* __restrict__ route,
                                                  const uint16_t* __restrict__ W2, int N, int I,
                                                  float* __restrict__ out, const VhXchg px) {
    __shared__ float red[4 * R];
    const int e0 = route[0], e1 = route[1];
    const float w0 = __int_as_float(route[2]), w1 = __int_as_float(route[3]);
    const int n0 = blockIdx.x * R;
    const uint16_t* rows0[R];
    const uint16_t* rows1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        rows0[r] = W2 + ((size_t)e0 * N + min(n0 + r, N - 1)) * I;
        rows1[r] = W2 + ((size_t)e1 * N + min(n0 + r, N - 1)) * I;
    }
    uint4 wa[R][NJ], wb[R][NJ];
    gemv_issue<NJ, R>(rows0, I, wa);
    gemv_issue<NJ, R>(rows1, I, wb);
    float tot[R];
    {
        float xr[NJ][8];
        load_x<NJ>(hbuf, I, xr);
        float acc[R];
        gemv_fma<NJ, R>(wa, xr, acc);
#pragma unroll
        for (int r = 0; r < R; ++r) tot[r] = w0 * acc[r];
    }
    {
        float xr[NJ][8];
        load_x<NJ>(hbuf + (size_t)I, I, xr);
        float acc[R];
        gemv_fma<NJ, R>(wb, xr, acc);
#pragma unroll
        for (int r = 0; r < R; ++r) tot[r] = fmaf(w1, acc[r], tot[r]);
    }
    // the thread sum is linear, so the routing weights were applied per thread above
    block256_sum<R>(tot, red);
    if (threadIdx.x < R && n0 + threadIdx.x < N) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) if (threadIdx.x == r) v = tot[r];
        if (px.world) xchg_put(px, n0 + threadIdx.x, v);
        else out[n0 + threadIdx.x] = v;
    }
}

// (r06 experiment, removed: gate|up and down of a tensor-parallel shard as ONE launch — down blocks behind the gate|up blocks of one grid,
// W2 rows requested once block 0 had published the route as granules, h read as tagged granules.  Bit-identical, and SLOWER: one rank's
// TP = 8 shard 1.369 against 1.313 ms per token, TP = 4 2.233 against 1.808 — 512-1024 resident down blocks poll the route and then h for
// ~10 us next to a gate|up phase that is pure streaming, and the 16-granule-per-thread sweeps of h cost more than the boundary and the W2
// round trip they replace: profiles/EXPERIMENTS.md.)

// ---- K_F: final RMSNorm + LM head GEMV + per-block argmax ---------------------------
#define LM_R 8
template <int NJ>
__global__ __launch_bounds__(256) void k_dec_lmhead(const float* __restrict__ x_in, const float* __restrict__ delta,
                                                    const float* __restrict__ norm_w, float eps,
                                                    const uint16_t* __restrict__ W, int V, int K,
                                                    float* __restrict__ logits, float* __restrict__ blk_val,
                                                    int* __restrict__ blk_idx, const int* __restrict__ ngen_ptr,
                                                    int hist_rows, int v0, int Vfull, const VhXchg cx) {
    __shared__ float red[4 * (LM_R + 1)];
    xchg_reduce(cx);
    __shared__ float bv_s[LM_R];
    __shared__ int bi_s[LM_R];
    // logits history: row = index of the token this step produces (clamped), so the host can
    // read every step's scores after a multi-step launch (HF generate(output_scores=True)).
    // W holds rows [v0, v0 + V) of the table (vocab-sharded head under tensor parallelism; v0 = 0, V = Vfull otherwise)
    if (hist_rows > 1) logits += (size_t)min(*ngen_ptr, hist_rows - 1) * Vfull;
    logits += v0;
    const int n_iter = (V + LM_R - 1) / LM_R;
    auto rows_of = [&](int it, const uint16_t* (&rows)[LM_R]) {
#pragma unroll
        for (int r = 0; r < LM_R; ++r) rows[r] = W + (size_t)min(it * LM_R + r, V - 1) * K;
    };
    int it = blockIdx.x;
    uint4 wa[LM_R][NJ];
    const uint16_t* rows[LM_R];
    if (it < n_iter) { rows_of(it, rows); gemv_issue<NJ, LM_R>(rows, K, wa); }
    float xr[NJ][8];
    const float ss = load_add_norm<NJ>(x_in, delta, norm_w, nullptr, K, xr, &cx);
    float inv = 0.f;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    bool first = true;
    while (it < n_iter) {
        float acc[LM_R];
        gemv_fma<NJ, LM_R>(wa, xr, acc);
        const int nxt = it + gridDim.x;
        if (nxt < n_iter) { rows_of(nxt, rows); gemv_issue<NJ, LM_R>(rows, K, wa); }
        float vals[LM_R + 1];
#pragma unroll
        for (int r = 0; r < LM_R; ++r) vals[r] = acc[r];
        vals[LM_R] = first ? ss : 0.f;
        block256_sum<LM_R + 1>(vals, red);
        if (first) { inv = rsqrtf(vals[LM_R] / (float)K + eps); first = false; }
        const int n0 = it * LM_R;
        if (threadIdx.x < LM_R && n0 + threadIdx.x < V) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < LM_R; ++r) if (threadIdx.x == r) v = vals[r];
            v *= inv;
            logits[n0 + threadIdx.x] = v;
            if (v > best) { best = v; besti = v0 + n0 + threadIdx.x; }  // ascending n: first max wins
        }
        it = nxt;
    }
    if (threadIdx.x < LM_R) { bv_s[threadIdx.x] = best; bi_s[threadIdx.x] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = bv_s[0]; int bi = bi_s[0];
        for (int r = 1; r < LM_R; ++r)
            if (bv_s[r] > b || (bv_s[r] == b && bi_s[r] < bi)) { b = bv_s[r]; bi = bi_s[r]; }
        blk_val[blockIdx.x] = b; blk_idx[blockIdx.x] = bi;
    }
}

// ---- K_G: global argmax (lowest index on ties, as torch.argmax), bookkeeping, and the
// next step's input embedding (vita_arch.py:155-175 decode early-exit + embed_tokens).
__global__ __launch_bounds__(256) void k_dec_select(const float* __restrict__ blk_val, const int* __restrict__ blk_idx,
                                                    int nblk, const uint16_t* __restrict__ embed, int H, int vocab,
                                                    float* __restrict__ x_next, int* __restrict__ pos_ptr,
                                                    int* __restrict__ ngen_ptr, int* __restrict__ out_tokens,
                                                    int max_out, int mode, int set_pos) {
    __shared__ float v_s[256];
    __shared__ int i_s[256];
    float b = -INFINITY; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        const float v = blk_val[i]; const int ix = blk_idx[i];
        if (v > b || (v == b && ix < bi)) { b = v; bi = ix; }
    }
    v_s[threadIdx.x] = b; i_s[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float v = v_s[threadIdx.x + s]; const int ix = i_s[threadIdx.x + s];
            if (v > v_s[threadIdx.x] || (v == v_s[threadIdx.x] && ix < i_s[threadIdx.x])) {
                v_s[threadIdx.x] = v; i_s[threadIdx.x] = ix;
            }
        }
        __syncthreads();
    }
    // all-NaN logits leave the sentinel index: never gather outside the table
    const int tok = (i_s[0] >= 0 && i_s[0] < vocab) ? i_s[0] : 0;
    if (threadIdx.x == 0) {
        // mode 1: decode step (append, pos++); mode 0: end of prefill (first token, pos = set_pos)
        const int n = mode ? *ngen_ptr : 0;
        if (n < max_out) out_tokens[n] = tok;
        *ngen_ptr = n + 1;
        *pos_ptr = mode ? (*pos_ptr + 1) : set_pos;
    }
    for (int c = threadIdx.x; c * 8 < H; c += 256) {
        const uint4 w = reinterpret_cast<const uint4*>(embed + (size_t)tok * H)[c];
        reinterpret_cast<float4*>(x_next)[c * 2] =
            make_float4(bf16_lo_to_f32(w.x), bf16_hi_to_f32(w.x), bf16_lo_to_f32(w.y), bf16_hi_to_f32(w.y));
        reinterpret_cast<float4*>(x_next)[c * 2 + 1] =
            make_float4(bf16_lo_to_f32(w.z), bf16_hi_to_f32(w.z), bf16_lo_to_f32(w.w), bf16_hi_to_f32(w.w));
    }
}

// ---- the global argmax alone (per-operator entry vh_lmhead_argmax): lowest index on ties, as torch.argmax ------------
__global__ __launch_bounds__(256) void k_dec_pick(const float* __restrict__ blk_val, const int* __restrict__ blk_idx, int nblk,
                                                  int vocab, int* __restrict__ token_out, float* __restrict__ value_out) {
    __shared__ float v_s[256];
    __shared__ int i_s[256];
    float b = -INFINITY; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        const float v = blk_val[i]; const int ix = blk_idx[i];
        if (v > b || (v == b && ix < bi)) { b = v; bi = ix; }
    }
    v_s[threadIdx.x] = b; i_s[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float v = v_s[threadIdx.x + s]; const int ix = i_s[threadIdx.x + s];
            if (v > v_s[threadIdx.x] || (v == v_s[threadIdx.x] && ix < i_s[threadIdx.x])) { v_s[threadIdx.x] = v; i_s[threadIdx.x] = ix; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        token_out[0] = (i_s[0] >= 0 && i_s[0] < vocab) ? i_s[0] : 0;
        if (value_out) value_out[0] = v_s[0];
    }
}

// ---- vocab-sharded LM head: this rank's best (value, index) -> its slot of a zeroed [world][2] vector; the
// all-reduce(sum) of that vector hands every rank every candidate (adding zeros is exact, indices < 2^24 are exact
// in fp32), k_dec_select then takes the global argmax with the usual lowest-index tie rule on every rank alike.
__global__ __launch_bounds__(256) void k_dec_cand(const float* __restrict__ blk_val, const int* __restrict__ blk_idx,
                                                  int nblk, float* __restrict__ cand, int rank, int world) {
    __shared__ float v_s[256];
    __shared__ int i_s[256];
    float b = -INFINITY; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        const float v = blk_val[i]; const int ix = blk_idx[i];
        if (v > b || (v == b && ix < bi)) { b = v; bi = ix; }
    }
    v_s[threadIdx.x] = b; i_s[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float v = v_s[threadIdx.x + s]; const int ix = i_s[threadIdx.x + s];
            if (v > v_s[threadIdx.x] || (v == v_s[threadIdx.x] && ix < i_s[threadIdx.x])) {
                v_s[threadIdx.x] = v; i_s[threadIdx.x] = ix;
            }
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < 2 * world) {
        float o = 0.f;
        if ((int)threadIdx.x == 2 * rank) o = v_s[0];
        if ((int)threadIdx.x == 2 * rank + 1) o = (float)i_s[0];
        cand[threadIdx.x] = o;
    }
}
__global__ void k_dec_cand_unpack(const float* __restrict__ cand, int world, float* __restrict__ val, int* __restrict__ idx) {
    if ((int)threadIdx.x < world) { val[threadIdx.x] = cand[2 * threadIdx.x]; idx[threadIdx.x] = (int)cand[2 * threadIdx.x + 1]; }
}


// ======================================================================================================================
// Batched decode: ONE iteration of up to VH_BMAX concurrent sequences (continuous batching over the paged KV cache,
// SURVEY 8(f)#1).  The weights are what a decode step streams (25.7 GB per token); a batch reads the attention-side
// weights and the LM head ONCE for all its sequences and every expert once per iteration however many sequences routed
// to it (expected 3.5 unique of 4 picks at B = 2, 5.5 of 8 at B = 4).  Per sequence the arithmetic is the batch-1
// kernels' (same per-thread accumulation order, same block reduction), so a sequence's ids do not depend on its
// neighbours.  Sequences are addressed through small by-value pointer tables; slots past `n` repeat the last sequence
// (their results are never stored) so every load is unconditional.
template <int NJ, int R, bool NORM>
__global__ __launch_bounds__(256) void k_decb_gemv(const VhDecBatchVec bt, const float* __restrict__ norm_w, float eps,
                                                   const uint16_t* __restrict__ W, int N, int K) {
    constexpr int NV = VH_BMAX * (R + 1);
    __shared__ float red[4 * NV];
    __shared__ float tot[NV];
    const int n0 = blockIdx.x * R;
    const uint16_t* rows[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rows[r] = W + (size_t)min(n0 + r, N - 1) * K;
    uint4 w[R][NJ];
    gemv_issue<NJ, R>(rows, K, w);
    float xr[VH_BMAX][NJ][8];
    float vals[VH_BMAX * (R + 1)];
#pragma unroll
    for (int b = 0; b < VH_BMAX; ++b) {
        const int bb = min(b, bt.n - 1);
        if (NORM) vals[b * (R + 1) + R] = load_add_norm<NJ>(bt.x_in[bb], bt.delta[bb], norm_w, b < bt.n ? bt.x_out[bb] : nullptr, K, xr[b]);
        else { load_x<NJ>(bt.x_in[bb], K, xr[b]); vals[b * (R + 1) + R] = 0.f; }
    }
#pragma unroll
    for (int b = 0; b < VH_BMAX; ++b) {
        float acc[R];
        gemv_fma<NJ, R>(w, xr[b], acc);
#pragma unroll
        for (int r = 0; r < R; ++r) vals[b * (R + 1) + r] = acc[r];
    }
    block256_multi_sum<NV>(vals, red, tot);      // 20 values: transposing reduction (wave_sum per value: 6 x 20 exchanges)
    if ((int)threadIdx.x < bt.n * R) {
        const int b = threadIdx.x / R, r = threadIdx.x - b * R;
        const float inv = NORM ? rsqrtf(tot[b * (R + 1) + R] / (float)K + eps) : 1.0f;
        float* o = bt.out[0];
#pragma unroll
        for (int q = 1; q < VH_BMAX; ++q) if (b == q) o = bt.out[q];
        if (n0 + r < N) o[n0 + r] = tot[b * (R + 1) + r] * inv;
    }
}

// attention of the batch: grid (nkv, max splits of the batch, n); every sequence has its own position, page table
// and partial / ticket buffers
__global__ __launch_bounds__(256) void k_decb_attn(const VhDecBatchAttn bt, float* __restrict__ kcache,
                                                   float* __restrict__ vcache, const float* __restrict__ rope_cos,
                                                   const float* __restrict__ rope_sin, int nq, int nkv, int max_ctx,
                                                   int max_splits, float scale) {
    const int b = blockIdx.z;
    const int nsplit = (bt.pos[b] + 1 + DA_KT - 1) / DA_KT;
    if ((int)blockIdx.y >= nsplit) return;
    DecAttnTile tile;
    dec_attn_issue(blockIdx.x, blockIdx.y, kcache, vcache, bt.table[b], max_ctx, tile);
    dec_attn_finish<false>(blockIdx.x, blockIdx.y, nsplit, tile, bt.qkv[b], kcache, vcache, bt.pos[b], bt.table[b], rope_cos, rope_sin,
                           bt.part_o[b], bt.part_ml[b], bt.cnt[b], bt.attn_out[b], nq, nkv, max_ctx, max_splits, scale, VhGranVec{}, VhGranVec{});
}

// final RMSNorm + LM head + per-block argmax for every sequence of the batch (the 424 MB table is read once)
template <int NJ>
__global__ __launch_bounds__(256) void k_decb_lmhead(const VhDecBatchVec bt, const float* __restrict__ norm_w, float eps,
                                                     const uint16_t* __restrict__ W, int V, int K, const VhDecBatchHead hd,
                                                     int v0) {
    constexpr int NV = VH_BMAX * (LM_R + 1);
    __shared__ float red[4 * NV];
    __shared__ float tot[NV];
    __shared__ float bv_s[VH_BMAX][LM_R];
    __shared__ int bi_s[VH_BMAX][LM_R];
    const int n_iter = (V + LM_R - 1) / LM_R;
    auto rows_of = [&](int it, const uint16_t* (&rows)[LM_R]) {
#pragma unroll
        for (int r = 0; r < LM_R; ++r) rows[r] = W + (size_t)min(it * LM_R + r, V - 1) * K;
    };
    int it = blockIdx.x;
    uint4 wa[LM_R][NJ];
    const uint16_t* rows[LM_R];
    if (it < n_iter) { rows_of(it, rows); gemv_issue<NJ, LM_R>(rows, K, wa); }
    float xr[VH_BMAX][NJ][8];
    float ss[VH_BMAX], inv[VH_BMAX], best[VH_BMAX];
    int besti[VH_BMAX];
#pragma unroll
    for (int b = 0; b < VH_BMAX; ++b) {
        const int bb = min(b, bt.n - 1);
        ss[b] = load_add_norm<NJ>(bt.x_in[bb], bt.delta[bb], norm_w, nullptr, K, xr[b]);
        inv[b] = 0.f; best[b] = -INFINITY; besti[b] = 0x7fffffff;
    }
    bool first = true;
    while (it < n_iter) {
        float vals[VH_BMAX * (LM_R + 1)];
#pragma unroll
        for (int b = 0; b < VH_BMAX; ++b) {
            float acc[LM_R];
            gemv_fma<NJ, LM_R>(wa, xr[b], acc);
#pragma unroll
            for (int r = 0; r < LM_R; ++r) vals[b * (LM_R + 1) + r] = acc[r];
            vals[b * (LM_R + 1) + LM_R] = first ? ss[b] : 0.f;
        }
        const int nxt = it + gridDim.x;
        if (nxt < n_iter) { rows_of(nxt, rows); gemv_issue<NJ, LM_R>(rows, K, wa); }
        block256_multi_sum<NV>(vals, red, tot);
        const int n0 = it * LM_R;
#pragma unroll
        for (int b = 0; b < VH_BMAX; ++b) {
            if (first) inv[b] = rsqrtf(tot[b * (LM_R + 1) + LM_R] / (float)K + eps);
            if (b < bt.n && threadIdx.x < LM_R && n0 + threadIdx.x < V) {
                float v = tot[b * (LM_R + 1) + threadIdx.x];
                v *= inv[b];
                if (hd.logits[b]) hd.logits[b][v0 + n0 + threadIdx.x] = v;
                if (v > best[b]) { best[b] = v; besti[b] = v0 + n0 + threadIdx.x; }
            }
        }
        first = false;
        it = nxt;
    }
#pragma unroll
    for (int b = 0; b < VH_BMAX; ++b)
        if (threadIdx.x < LM_R) { bv_s[b][threadIdx.x] = best[b]; bi_s[b][threadIdx.x] = besti[b]; }
    __syncthreads();
    if ((int)threadIdx.x < bt.n) {
        const int b = threadIdx.x;
        float bv = bv_s[b][0]; int bi = bi_s[b][0];
        for (int r = 1; r < LM_R; ++r)
            if (bv_s[b][r] > bv || (bv_s[b][r] == bv && bi_s[b][r] < bi)) { bv = bv_s[b][r]; bi = bi_s[b][r]; }
        hd.blk_val[b][blockIdx.x] = bv; hd.blk_idx[b][blockIdx.x] = bi;
    }
}

template <typename F>
inline int pick_nj(int K, F&& f) {
    // chunk slots per thread: K <= NJ * 2048
    if (K <= 2048) return f(std::integral_constant<int, 1>{});
    if (K <= 4096) return f(std::integral_constant<int, 2>{});
    if (K <= 8192) return f(std::integral_constant<int, 4>{});
    if (K <= 14336) return f(std::integral_constant<int, 7>{});
    return -1;
}

}  // namespace

// ---- launchers (declared in vh_kernels.h) -------------------------------------------
static inline VhXchg xchg_or_none(const VhXchg* x) {
    VhXchg z{};
    return x ? *x : z;            // world == 0: no exchange
}

template <int R, bool NORM>
static int launch_dec_gemv(hipStream_t st, const float* x_in, const float* delta, float* x_out, const float* norm_w,
                           float eps, const uint16_t* W, int N, int K, float* out, const VhXchg* xc) {
    return pick_nj(K, [&](auto nj) {
        constexpr int NJ = decltype(nj)::value;
        hipLaunchKernelGGL((k_dec_gemv<NJ, R, NORM>), dim3((N + R - 1) / R), dim3(256), 0, st, x_in, delta, x_out, norm_w, eps, W, N, K, out,
                           xchg_or_none(xc));
        return 0;
    });
}

// rows per block of the QKV / O GEMVs.  8: r02 sweep with the transposing block reduction (4: -0.5 % of a token, 16: -1 %)
constexpr int DEC_GEMV_R = 8;
// (r06: groups of 7 + 7 rows for tensor-parallel shards — 512 groups = ONE round on 512 blocks at I = 1792 instead of 2-3 rounds of 4 + 4 —
// measured SLOWER at every degree: one rank's TP = 8 / 4 / 2 shard 1.316 / 1.831 / 2.871 against 1.287 / 1.788 / 2.790 ms per token; removed)
static int dec_gateup_grid(int I) {
    constexpr int rp = 4;
    // 1.5 persistent blocks per CU: every block pays the router prologue (96 KB of L2 reads), so fewer, longer-lived
    // blocks win — r01: 79 us at 512 blocks vs 83 at 1024 and 82 at 1280; r02 (cheaper block reductions): whole-token rate
    // 196.6 / 205.7 / 210.5 / 207.6 / 207.0 / 208.7 / 207.4 tok/s at 256 / 320 / 384 / 448 / 512 / 768 / 1024 blocks
    const int n_iter = 2 * (I / rp);
    const int cus = vh_num_cus();
    int grid = 3 * cus / 2;
    if (vh_tuning()->dec_gateup_grid > 0) grid = vh_tuning()->dec_gateup_grid;
    else if (n_iter <= 4 * grid) {
        // (r06) the few row groups of a tensor-parallel shard: EQUAL shares in as few rounds as 2 blocks per CU allow
        const int rounds = (n_iter + 2 * cus - 1) / (2 * cus);
        grid = (n_iter + rounds - 1) / rounds;
    }
    return grid > n_iter ? n_iter : grid;
}
// blocks of a consumer launch: the fused exchange's reducers are its first min(16, blocks) blocks
int vhk_dec_consumer_blocks(int which, int N, int K, int I) {
    (void)K;
    if (which == 0) return (N + DEC_GEMV_R - 1) / DEC_GEMV_R;
    if (which == 1) return dec_gateup_grid(I);
    if (which == 3) return N;   // fused attention block: at least the fused-QKV blocks (N = their count) come first
    return N;   // LM head: the caller's grid
}

int vhk_dec_qkv(hipStream_t st, const float* x_in, const float* delta, float* x_out, const float* norm_w, float eps,
                const uint16_t* W, int N, int K, float* out, const VhXchg* cx) {
    return launch_dec_gemv<DEC_GEMV_R, true>(st, x_in, delta, x_out, norm_w, eps, W, N, K, out, cx);
}

int vhk_dec_ablk_supported(int H, int nq, int nkv) {
    return H <= 4096 && H % 8 == 0 && nq * 128 <= 4096 && nkv >= 1 && nq % nkv == 0 && nq / nkv <= 4;
}
// rows per fused-QKV block: 8 (the per-operator kernel's), fewer for a tensor-parallel shard's short matrix so that its blocks still
// cover the chip (768 rows at TP = 8: 384 blocks of 2)
static int dec_ablk_rq(int nqkv, int H) {
    if (H <= 2048) return 8;                                          // one chunk slot per thread: a single instantiation
    const int cus = vh_num_cus();
    for (int r : {8, 4, 2}) if ((nqkv + r - 1) / r >= cus + cus / 2) return r;
    return 2;
}
int vhk_dec_ablk_qkv_blocks(int nqkv, int H) { const int r = dec_ablk_rq(nqkv, H); return (nqkv + r - 1) / r; }
int vhk_dec_ablk(hipStream_t st, const VhDecAblk& a) {
    if (!vhk_dec_ablk_supported(a.H, a.nq, a.nkv) || a.nsplit < 1 || a.nsplit > a.max_splits || !a.gq.g || !a.ga.g) return -1;
    const int KO = a.nq * 128;
    const int rq = dec_ablk_rq(a.nqkv, a.H);
    const long total = (long)(a.nqkv + rq - 1) / rq + (long)a.nkv * a.nsplit + (a.H + 7) / 8;
    if (total > 1000000) return -1;
    const dim3 grid((unsigned)total);
    auto launch = [&](auto nj, auto njo, auto r) {
        hipLaunchKernelGGL((k_dec_ablk<decltype(nj)::value, decltype(njo)::value, decltype(r)::value>), grid, dim3(256), 0, st, a);
        return 0;
    };
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    auto by_o = [&](auto nj, auto r) { return KO <= 2048 ? launch(nj, I1{}, r) : launch(nj, I2{}, r); };
    if (a.H <= 2048) return by_o(I1{}, std::integral_constant<int, 8>{});
    switch (rq) {
        case 2: return by_o(I2{}, std::integral_constant<int, 2>{});
        case 4: return by_o(I2{}, std::integral_constant<int, 4>{});
        default: return by_o(I2{}, std::integral_constant<int, 8>{});
    }
}

int vhk_dec_attn(hipStream_t st, const float* qkv, float* kcache, float* vcache, const int* pos_ptr,
                 const float* rope_cos, const float* rope_sin, float* part_o, float* part_ml, int* cnt,
                 float* attn_out, int nq, int nkv, int max_ctx, int max_splits, int ctx_host, float scale,
                 const int* table) {
    (void)pos_ptr;  // the kernel takes the position from the host mirror (ctx_host - 1)
    if (nq % nkv != 0 || nq / nkv > 4) return -1;
    const int nsplit = (ctx_host + DA_KT - 1) / DA_KT;
    if (nsplit < 1 || nsplit > max_splits || nsplit > 65535) return -1;
    hipLaunchKernelGGL(k_dec_attn, dim3(nkv, nsplit), dim3(256), 0, st, qkv, kcache, vcache, ctx_host - 1, table, rope_cos,
                       rope_sin, part_o, part_ml, cnt, attn_out, nq, nkv, max_ctx, max_splits, scale);
    return 0;
}

int vhk_dec_oproj(hipStream_t st, const float* attn_out, const uint16_t* W, int N, int K, float* out, const VhXchg* px) {
    return launch_dec_gemv<DEC_GEMV_R, false>(st, attn_out, nullptr, nullptr, nullptr, 0.f, W, N, K, out, px);
}

int vhk_dec_gateup(hipStream_t st, const float* x_in, const float* delta, float* x_out, const float* norm_w, float eps,
                   const uint16_t* Wg, int E, const uint16_t* W1, const uint16_t* W3, int I, int K, int* route_out,
                   float* hbuf, int grid, const VhXchg* cxp) {
    if (E > 8 || E < 2 || I % 4 != 0) return -1;
    const int n_iter = 2 * (I / 4);
    const VhXchg cx = xchg_or_none(cxp);
    if (grid <= 0) grid = dec_gateup_grid(I);
    if (grid > n_iter) grid = n_iter;
    return pick_nj(K, [&](auto nj) {
        constexpr int NJ = decltype(nj)::value;
        hipLaunchKernelGGL((k_dec_gateup<NJ, 4>), dim3(grid), dim3(256), 0, st, x_in, delta, x_out, norm_w, eps, Wg, E, W1, W3, I, K,
                           route_out, hbuf, cx);
        return 0;
    });
}

int vhk_dec_down(hipStream_t st, const float* hbuf, const int* route, const uint16_t* W2, int N, int I, float* out,
                 const VhXchg* pxp) {
    // one block per row pair (a persistent form that keeps both intermediate vectors in registers measured slower: 204.6-208.8
    // against 211.5 tok/s, r02, git history)
    constexpr int R = 2;
    const VhXchg px = xchg_or_none(pxp);
    return pick_nj(I, [&](auto nj) {
        hipLaunchKernelGGL((k_dec_down<decltype(nj)::value, R>), dim3((N + R - 1) / R), dim3(256), 0, st, hbuf, route,
                           W2, N, I, out, px);
        return 0;
    });
}

int vhk_dec_lmhead(hipStream_t st, const float* x_in, const float* delta, const float* norm_w, float eps,
                   const uint16_t* W, int V, int K, float* logits, float* blk_val, int* blk_idx, int grid,
                   const int* ngen_ptr, int hist_rows, int v0, int Vfull, const VhXchg* cx) {
    return pick_nj(K, [&](auto nj) {
        hipLaunchKernelGGL((k_dec_lmhead<decltype(nj)::value>), dim3(grid), dim3(256), 0, st, x_in, delta, norm_w, eps,
                           W, V, K, logits, blk_val, blk_idx, ngen_ptr, hist_rows, v0, Vfull, xchg_or_none(cx));
        return 0;
    });
}
int vhk_dec_cand(hipStream_t st, const float* blk_val, const int* blk_idx, int nblk, float* cand, int rank, int world) {
    if (world < 1 || world > 128) return -1;
    hipLaunchKernelGGL(k_dec_cand, dim3(1), dim3(256), 0, st, blk_val, blk_idx, nblk, cand, rank, world);
    return 0;
}
int vhk_dec_cand_unpack(hipStream_t st, const float* cand, int world, float* val, int* idx) {
    hipLaunchKernelGGL(k_dec_cand_unpack, dim3(1), dim3(128), 0, st, cand, world, val, idx);
    return 0;
}

int vhk_dec_pick(hipStream_t st, const float* blk_val, const int* blk_idx, int nblk, int vocab, int* token_out, float* value_out) {
    if (nblk < 1) return -1;
    hipLaunchKernelGGL(k_dec_pick, dim3(1), dim3(256), 0, st, blk_val, blk_idx, nblk, vocab, token_out, value_out);
    return 0;
}

int vhk_dec_select(hipStream_t st, const float* blk_val, const int* blk_idx, int nblk, const uint16_t* embed, int H,
                   int vocab, float* x_next, int* pos_ptr, int* ngen_ptr, int* out_tokens, int max_out, int mode, int set_pos) {
    hipLaunchKernelGGL(k_dec_select, dim3(1), dim3(256), 0, st, blk_val, blk_idx, nblk, embed, H, vocab, x_next, pos_ptr,
                       ngen_ptr, out_tokens, max_out, mode, set_pos);
    return 0;
}

// ---- batched decode launchers ---------------------------------------------------------------------------------------
int vhk_decb_gemv(hipStream_t st, const VhDecBatchVec& bt, const float* norm_w, float eps, const uint16_t* W, int N, int K,
                  int norm) {
    if (bt.n < 1 || bt.n > VH_BMAX) return -1;
    constexpr int R = 4;
    return pick_nj(K, [&](auto nj) {
        constexpr int NJ = decltype(nj)::value;
        if (NJ > 2) return -1;                       // batch slices of the activation live in registers: K <= 4096
        if (norm) hipLaunchKernelGGL((k_decb_gemv<(NJ > 2 ? 2 : NJ), R, true>), dim3((N + R - 1) / R), dim3(256), 0, st, bt, norm_w, eps, W, N, K);
        else hipLaunchKernelGGL((k_decb_gemv<(NJ > 2 ? 2 : NJ), R, false>), dim3((N + R - 1) / R), dim3(256), 0, st, bt, norm_w, eps, W, N, K);
        return 0;
    });
}
int vhk_decb_attn(hipStream_t st, const VhDecBatchAttn& bt, int n, float* kcache, float* vcache, const float* rope_cos,
                  const float* rope_sin, int nq, int nkv, int max_ctx, int max_splits, float scale) {
    if (n < 1 || n > VH_BMAX || nq % nkv != 0 || nq / nkv > 4) return -1;
    int ms = 1;
    for (int b = 0; b < n; ++b) {
        const int ns = (bt.pos[b] + 1 + DA_KT - 1) / DA_KT;
        if (ns < 1 || ns > max_splits) return -1;
        if (ns > ms) ms = ns;
    }
    hipLaunchKernelGGL(k_decb_attn, dim3(nkv, ms, n), dim3(256), 0, st, bt, kcache, vcache, rope_cos, rope_sin, nq, nkv,
                       max_ctx, max_splits, scale);
    return 0;
}
int vhk_decb_lmhead(hipStream_t st, const VhDecBatchVec& bt, const float* norm_w, float eps, const uint16_t* W, int V, int K,
                    const VhDecBatchHead& hd, int grid, int v0) {
    if (bt.n < 1 || bt.n > VH_BMAX) return -1;
    return pick_nj(K, [&](auto nj) {
        constexpr int NJ = decltype(nj)::value;
        if (NJ > 2) return -1;
        hipLaunchKernelGGL((k_decb_lmhead<(NJ > 2 ? 2 : NJ)>), dim3(grid), dim3(256), 0, st, bt, norm_w, eps, W, V, K, hd, v0);
        return 0;
    });
}
