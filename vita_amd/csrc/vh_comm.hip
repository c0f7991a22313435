// vh_comm.hip — the library's own all-reduce(sum) over IPC-mapped peer buffers (the exchange after
// o_proj and after the MoE down projection — web_demo/vllm_tools/vllm_file/mixtral.py:405-414 FusedMoE reduce_results,
// :470-476 RowParallel o_proj).
//
// Why not only RCCL.  A decode step carries 64 all-reduces of ONE 16 KB vector; a ring / tree collective pays several
// hops of flag latency per call.  On an MI355X node every GPU has a direct xGMI link to each of its 7 peers, so the
// latency-optimal exchange is direct: every rank PUSHES its vector into a slot of every peer's receive buffer and then
// sums the world slots it finds in its OWN memory, in rank order (so all ranks hold bit-identical sums — the replicated
// router / norm computations downstream must not diverge).
//
// Transport: 8-byte granules {value : fp32 bits, tag : epoch} written with ONE system-scope store each, polled with
// system-scope loads until the tag matches (the data is the flag: no fence, no separate flag word — guide §6 G16 R2 /
// NCCL "LL").  Receive buffers are allocated fine-grained (uncached) by the library so that a peer's stores become
// visible to a spinning kernel; they are double-buffered by epoch parity: rank A can only overwrite the slot of epoch
// e at epoch e+2, which it reaches only after receiving B's epoch e+1 data, i.e. after B finished reading epoch e.
//   one-shot  (count <= VH_COMM_ONESHOT_MAX): 7 remote stores + 8 local polls per element, one exchange.
//   bulk      (larger, the prefill messages; r04): reduce-scatter — rank r pushes slice s of its vector to slice owner s,
//             the owner sums the world contributions in rank order — then all-gather — the owner pushes the reduced
//             slice to every rank: 2 (W-1)/W N elements per rank spread over all 7 links at once (a ring moves the same
//             over ONE link per direction).  At these sizes (4.5 / 9 MB) the links' BYTES are the cost, so the payload
//             travels as plain fp32 in 16-KB chunks with ONE flag per chunk (16-byte stores -> every wave drains ->
//             one system-scope release -> relaxed flag store; the reader polls the flag, one system-scope acquire, plain
//             loads: guide G16 R1) — half the bytes of r01-r03's granule form of the same schedule, which paid 8 bytes
//             per value to save a flag round trip that does not matter here.
// Every spin is bounded; a time-out sets the error word read by vh_comm_status().  Bring-up (vita_amd/parallel.py)
// self-tests the path against torch.distributed before it is used and falls back to RCCL otherwise.
#include <stdio.h>
#include <string.h>

#include "../../include/vita_hip.h"
#include "vh_common.h"
#include "vh_kernels.h"

#define VH_COMM_MAX_WORLD 8
#define VH_COMM_ONESHOT_MAX 32768          // elements: 256 KB of granules per peer slot
#define VH_COMM_SPIN_LIMIT (1u << 26)      // polls (with s_sleep): seconds, never a device hang

struct vh_comm {
    int rank, world;
    size_t cap;                              // fp32 elements per all-reduce
    size_t region;                           // 8-byte units per parity region: one-shot granule slots, then the bulk area
    size_t oneshot_units;                    // size of the one-shot part (even)
    int maxchunk;                            // flag words per rank row of the bulk area
    uint64_t* local;                         // [2 parity][region] granules, fine-grained
    uint64_t* peer[VH_COMM_MAX_WORLD];       // every rank's buffer as mapped here (peer[rank] == local)
    bool opened[VH_COMM_MAX_WORLD];
    int* err;                                // device error word
    uint64_t calls;                          // all-reduces issued (every rank counts the same calls: the API is collective)
    uint32_t generation;                     // tag wraps survived (barrier tags)
    bool connected;
    bool fine_grained;                       // the receive buffer is uncached / coherent for peer stores
    int ranks_per_device;                    // declared at creation (vh_tune("comm_ranks_per_device", n)): ranks that drive THIS rank's device
    bool loopback;                           // one rank plays all `world` ranks into its own slots (vh_comm_create_loopback)
    // exchanges fused into the decode kernels (VhXchg): two result vectors (attention / MoE sub-block) as tagged granules in the GEMV layout
    unsigned long long* reduced_g[2];
};
// buffer layout (granules): [2 parity regions][2 barrier rows of VH_COMM_MAX_WORLD].  The 32-bit tag of a call is the low
// word of the call counter (0 is skipped: "never written"); the parity region is the counter's low bit, tracked
// SEPARATELY from the tag, so consecutive calls always alternate regions even across a tag wrap.  When the tag wraps
// (2^32 calls: days of serving) every rank runs the same re-initialisation at the same call: barrier, zero both regions,
// barrier — a granule left from 2^32 calls ago can then never match a new tag.

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ void put(uint64_t* p, uint32_t tag, float v) {
    __hip_atomic_store(reinterpret_cast<u64*>(p), ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
// poll one granule until its tag is `tag`; returns the value (0 and *err = code on time-out)
// A time-out (dead peer) is sticky: once the error word is set every later poll of every kernel gives up after ONE look,
// so a lost peer costs one spin limit in all, not one per granule.
__device__ __forceinline__ float get(const uint64_t* p, uint32_t tag, int* err, int code) {
    unsigned spins = 0;
    for (;;) {
        const u64 x = __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((uint32_t)(x >> 32) == tag) return __uint_as_float((uint32_t)x);
        if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return 0.f;
        if (++spins > VH_COMM_SPIN_LIMIT) {
            __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return 0.f;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

struct Peers { uint64_t* p[VH_COMM_MAX_WORLD]; };

// region layout (granules): one-shot: slot r at [r * cap, ...)
__global__ __launch_bounds__(256) void k_ar_oneshot(float* __restrict__ buf, long count, Peers peers, uint64_t* local,
                                                    size_t cap, int rank, int world, uint32_t tag, int* err, int loopback) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float v = buf[i];
        // loopback: this rank plays all of them — its value into slot `rank`, the peers' (zero) contributions into theirs
        if (loopback) for (int p = 0; p < world; ++p) put(local + (size_t)p * cap + i, tag, p == rank ? v : 0.f);
        else for (int p = 0; p < world; ++p) put(peers.p[p] + (size_t)rank * cap + i, tag, v);
        // all `world` slots of the element polled TOGETHER (r01-r05: one after the other — `world` dependent round trips to the
        // uncached receive buffer); a look that finds a tag missing re-reads every slot after a short sleep
        u64 g[VH_COMM_MAX_WORLD];
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int r = 0; r < VH_COMM_MAX_WORLD; ++r)
                if (r < world) g[r] = __hip_atomic_load(reinterpret_cast<const u64*>(local + (size_t)r * cap + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
            for (int r = 0; r < VH_COMM_MAX_WORLD; ++r)
                if (r < world) ok = ok && ((uint32_t)(g[r] >> 32) == tag);
            if (ok) break;
            if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (++spins > VH_COMM_SPIN_LIMIT) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < VH_COMM_MAX_WORLD; ++r)
            if (r < world) s += __uint_as_float((uint32_t)g[r]);   // rank order: same sum everywhere
        buf[i] = s;
    }
}

// ---- bulk path ---------------------------------------------------------------------------------------------------------
// slice = elements owned by one rank (multiple of 4), cut into nchunk chunks of VH_COMM_CHUNK elements (the last one short).
// The bulk area of a parity region starts BEHIND the one-shot slots (a payload word can never be read as a granule), and its
// flag words sit at FIXED positions in front of the payload (a flag word only ever holds flags: tags of earlier calls), in
// 8-byte units from the region start: FA = flags [src rank][maxchunk] at offFA ("rank r's contribution to chunk k of MY
// slice is complete"), FB = flags [owner][maxchunk] at offFB ("owner s's reduced chunk k is complete"), A = contributions
// [src rank][slice_pad] fp32 at offA, B = reduced slices [owner][slice_pad] fp32 at offB.  A flag holds the call's tag.
#define VH_COMM_CHUNK 4096
struct BulkGeom { long slice, slice_pad; int nchunk, maxchunk; size_t offFA, offFB, offA, offB; };

__device__ __forceinline__ void flag_set(uint64_t* f, uint32_t tag) {
    __hip_atomic_store(reinterpret_cast<u64*>(f), (u64)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void flag_wait(const uint64_t* f, uint32_t tag, int* err, int code) {
    unsigned spins = 0;
    for (;;) {
        if ((uint32_t)__hip_atomic_load(reinterpret_cast<const u64*>(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == tag) return;
        if ((spins & 1023u) == 1023u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        if (++spins > VH_COMM_SPIN_LIMIT) { __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        __builtin_amdgcn_s_sleep(2);
    }
}
// every storing wave has drained, then ONE lane releases at system scope and raises the flag(s) (guide G16 R1 / pitfall 12:
// the asm wait after the fence is what the compiler cannot drop)
__device__ __forceinline__ void publish_begin() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
__device__ __forceinline__ void consume_begin() {      // after the polling lanes saw their flags
    __syncthreads();
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
}
// n fp32 from src to dst, both 16-byte aligned at element 0 (slices and chunks are multiples of 4 elements)
__device__ __forceinline__ void copy_f32(float* __restrict__ dst, const float* __restrict__ src, long n, bool vec) {
    if (vec) {
        const long n4 = n >> 2;
        for (long i = threadIdx.x; i < n4; i += blockDim.x) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    } else {
        for (long i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(256) void k_ar_bulk(float* __restrict__ buf, long count, Peers peers, uint64_t* local, BulkGeom g,
                                                 int rank, int world, uint32_t tag, int* err, int vec) {
    const int items = world * g.nchunk;
    auto live = [&](int s, int k) { return (long)s * g.slice + (long)k * VH_COMM_CHUNK < min(count, (long)(s + 1) * g.slice); };
    auto len = [&](int s, int k) { return min(min(count, (long)(s + 1) * g.slice) - ((long)s * g.slice + (long)k * VH_COMM_CHUNK), (long)VH_COMM_CHUNK); };
    // 1. my contribution to every chunk of every slice -> its owner (consecutive blocks address different owners / links)
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int s = item % world, k = item / world;
        if (!live(s, k)) continue;                                   // block-uniform
        float* dst = reinterpret_cast<float*>(peers.p[s] + g.offA) + (size_t)rank * g.slice_pad + (size_t)k * VH_COMM_CHUNK;
        copy_f32(dst, buf + (size_t)s * g.slice + (size_t)k * VH_COMM_CHUNK, len(s, k), vec != 0);
        publish_begin();
        if (threadIdx.x == 0) flag_set(peers.p[s] + g.offFA + (size_t)rank * g.maxchunk + k, tag);
    }
    // 2. the chunks of MY slice: sum the world contributions in rank order, push the result to every rank
    for (int k = blockIdx.x; k < g.nchunk; k += gridDim.x) {
        if (!live(rank, k)) continue;
        const long n = len(rank, k);
        if ((int)threadIdx.x < world) flag_wait(local + g.offFA + (size_t)threadIdx.x * g.maxchunk + k, tag, err, 2);
        consume_begin();
        const float* A = reinterpret_cast<const float*>(local + g.offA) + (size_t)k * VH_COMM_CHUNK;
        const size_t bo = (size_t)rank * g.slice_pad + (size_t)k * VH_COMM_CHUNK;
        for (long i = (long)threadIdx.x * 4; i < n; i += (long)blockDim.x * 4) {
            if (i + 4 <= n) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int r = 0; r < world; ++r) acc += *reinterpret_cast<const f32x4*>(A + (size_t)r * g.slice_pad + i);
                for (int p = 0; p < world; ++p)
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(peers.p[p] + g.offB) + bo + i) = acc;
            } else {
                for (long j = i; j < n; ++j) {
                    float acc = 0.f;
                    for (int r = 0; r < world; ++r) acc += A[(size_t)r * g.slice_pad + j];
                    for (int p = 0; p < world; ++p) (reinterpret_cast<float*>(peers.p[p] + g.offB) + bo)[j] = acc;
                }
            }
        }
        publish_begin();
        if ((int)threadIdx.x < world) {
            // (lane 0 released; the other flag-writing lanes of wave 0 run after it in program order of the same wave)
            flag_set(peers.p[threadIdx.x] + g.offFB + (size_t)rank * g.maxchunk + k, tag);
        }
    }
    // 3. collect every reduced chunk
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int s = item % world, k = item / world;
        if (!live(s, k)) continue;
        if (threadIdx.x == 0) flag_wait(local + g.offFB + (size_t)s * g.maxchunk + k, tag, err, 3);
        consume_begin();
        copy_f32(buf + (size_t)s * g.slice + (size_t)k * VH_COMM_CHUNK,
                 reinterpret_cast<const float*>(local + g.offB) + (size_t)s * g.slice_pad + (size_t)k * VH_COMM_CHUNK, len(s, k), vec != 0);
    }
}

// all ranks have reached this point of their streams: rank r raises row[r] in every peer's barrier row, then waits for all
__global__ void k_comm_barrier(Peers bars, uint64_t* local_bar, int rank, int world, uint32_t tag, int* err) {
    const int t = threadIdx.x;
    if (t < world) put(bars.p[t] + rank, tag, 0.f);
    if (t < world) (void)get(local_bar + t, tag, err, 4);
}

thread_local char g_cerr[256] = "";
int cfail(int code, const char* msg, hipError_t e = hipSuccess) {
    if (e != hipSuccess) snprintf(g_cerr, sizeof(g_cerr), "%s: %s", msg, hipGetErrorString(e));
    else snprintf(g_cerr, sizeof(g_cerr), "%s", msg);
    return code;
}

// tag wrap: barrier (everybody finished every earlier all-reduce) -> zero both parity regions -> barrier (nobody pushes a
// new tag into a region that is not zeroed yet).  Barrier rows use the generation counter as their tag.
int comm_rewind(vh_comm* c, hipStream_t st) {
    for (int phase = 0; phase < 2; ++phase) {
        c->generation += 1;
        Peers bars{};
        for (int r = 0; r < c->world; ++r) bars.p[r] = c->peer[r] + 2 * c->region + (size_t)phase * VH_COMM_MAX_WORLD;
        hipLaunchKernelGGL(k_comm_barrier, dim3(1), dim3(64), 0, st, bars, c->local + 2 * c->region + (size_t)phase * VH_COMM_MAX_WORLD,
                           c->rank, c->world, c->generation, c->err);
        if (phase == 0 && (hipMemsetAsync(c->local, 0, 2 * c->region * sizeof(uint64_t), st) != hipSuccess ||
                           hipMemsetAsync(c->reduced_g[0], 0, (size_t)(c->reduced_g[1] - c->reduced_g[0]) * 2 * sizeof(unsigned long long), st) != hipSuccess))
            return cfail(VH_E_HIP, "vh_comm: re-zero at the tag wrap failed");
    }
    return VH_OK;
}

}  // namespace

extern "C" {

const char* vh_comm_last_error(void) { return g_cerr; }

// tests: continue from a given call count (e.g. 2^32 - 3 to cross the tag wrap); every rank must set the same value
int vh_comm_debug_set_calls(vh_comm_t* c, uint64_t calls) {
    if (!c) return VH_E_ARG;
    c->calls = calls;
    return VH_OK;
}
int vh_comm_is_fine_grained(const vh_comm_t* c) { return c && c->fine_grained ? 1 : 0; }

vh_comm_t* vh_comm_create(int rank, int world, size_t cap_elems, void* handle_out) {
    if (rank < 0 || world < 2 || world > VH_COMM_MAX_WORLD || rank >= world || cap_elems == 0 || !handle_out) {
        cfail(VH_E_ARG, "vh_comm_create: bad arguments (2 <= world <= 8)");
        return nullptr;
    }
    vh_comm* c = new vh_comm{};
    c->rank = rank; c->world = world; c->cap = cap_elems;
    c->ranks_per_device = vh_tuning()->comm_ranks_per_device;
    if (c->ranks_per_device < 1) c->ranks_per_device = 1;
    if (c->ranks_per_device > world) c->ranks_per_device = world;
    // one-shot needs world * cap granules; bulk (8-byte units): A and B of world * slice_pad fp32 each + two flag arrays
    c->oneshot_units = ((size_t)world * (cap_elems < VH_COMM_ONESHOT_MAX ? cap_elems : VH_COMM_ONESHOT_MAX) + 1) & ~size_t(1);
    c->maxchunk = (int)(cap_elems / ((size_t)world * VH_COMM_CHUNK)) + 2;
    const size_t pad_elems = cap_elems + (size_t)world * (VH_COMM_CHUNK + 4);                 // world * slice_pad at count = cap
    const size_t bulk = cap_elems > VH_COMM_ONESHOT_MAX ? 2 * (size_t)world * c->maxchunk + pad_elems + 2 : 0;
    c->region = (c->oneshot_units + bulk + 1) & ~size_t(1);       // even: both parity regions start on 16-byte boundaries
    const size_t bytes = (2 * c->region + 2 * VH_COMM_MAX_WORLD) * sizeof(uint64_t);
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->local), bytes, hipDeviceMallocFinegrained);
    c->fine_grained = e == hipSuccess;
    if (e != hipSuccess) {
        // A coarse-grained buffer is only correct when every rank sits on THIS device (same-device tests): across
        // devices a peer's stores may never become visible to a spinning kernel — refuse instead of timing out later.
        (void)hipGetLastError();
        if (!vh_tuning()->comm_allow_coarse) {
            cfail(VH_E_COMM, "vh_comm_create: fine-grained (peer-coherent) device memory is not available and the ranks are "
                             "not declared to share one device (vh_tune(\"comm_allow_coarse\", 1))", e);
            delete c;
            return nullptr;
        }
        e = hipMalloc(reinterpret_cast<void**>(&c->local), bytes);
    }
    if (e != hipSuccess) { cfail(VH_E_HIP, "vh_comm_create: buffer allocation", e); delete c; return nullptr; }
    const size_t red_elems = cap_elems < VH_COMM_ONESHOT_MAX ? cap_elems : VH_COMM_ONESHOT_MAX;
    if ((e = hipMemset(c->local, 0, bytes)) != hipSuccess || (e = hipMalloc(reinterpret_cast<void**>(&c->err), 2 * sizeof(int))) != hipSuccess ||
        (e = hipMemset(c->err, 0, 2 * sizeof(int))) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void**>(&c->reduced_g[0]), 2 * vh_gran_gemv_len((int)red_elems) * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMemset(c->reduced_g[0], 0, 2 * vh_gran_gemv_len((int)red_elems) * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipDeviceSynchronize()) != hipSuccess) {
        cfail(VH_E_HIP, "vh_comm_create: initialisation", e);
        (void)hipFree(c->local); (void)hipFree(c->err); (void)hipFree(c->reduced_g[0]); delete c; return nullptr;   // (null pointers are no-ops)
    }
    hipIpcMemHandle_t h;
    if ((e = hipIpcGetMemHandle(&h, c->local)) != hipSuccess) {
        cfail(VH_E_HIP, "vh_comm_create: hipIpcGetMemHandle", e);
        (void)hipFree(c->local); (void)hipFree(c->err); (void)hipFree(c->reduced_g[0]); delete c; return nullptr;
    }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    memcpy(handle_out, &h, sizeof(h));
    c->peer[rank] = c->local;
    c->calls = 0;
    c->generation = 0;
    c->reduced_g[1] = c->reduced_g[0] + vh_gran_gemv_len((int)red_elems);
    return c;
}

int vh_comm_connect(vh_comm_t* c, const void* handles) {
    if (!c || !handles) return cfail(VH_E_ARG, "vh_comm_connect: null argument");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char*>(handles) + (size_t)r * 64, 64);
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return cfail(VH_E_COMM, "vh_comm_connect: hipIpcOpenMemHandle", e);
        c->peer[r] = static_cast<uint64_t*>(p);
        c->opened[r] = true;
    }
    c->connected = true;
    return VH_OK;
}

/* A single-process communicator that plays `world` ranks (bench.py --emulate-tp N --loopback): every peer region is the local one,
 * a push writes this rank's value into slot `rank` and a zero into each of the other world - 1 slots, the reduction polls and sums
 * the world slots in rank order — the stores, polls, tags, parity regions and launches of a real exchange with no link in between,
 * and sums that equal the inputs (the tokens stay those of this rank's shard alone).  One-shot (decode-sized) messages only. */
vh_comm_t* vh_comm_create_loopback(int rank, int world, size_t cap_elems) {
    if (cap_elems == 0 || cap_elems > VH_COMM_ONESHOT_MAX) { cfail(VH_E_ARG, "vh_comm_create_loopback: capacity outside (0, 32768]"); return nullptr; }
    char handle[64];
    vh_comm* c = vh_comm_create(rank, world, cap_elems, handle);
    if (!c) return nullptr;
    for (int r = 0; r < world; ++r) c->peer[r] = c->local;
    c->loopback = true;
    c->connected = true;
    return c;
}

size_t vh_comm_capacity(const vh_comm_t* c) { return c ? c->cap : 0; }

int vh_comm_allreduce(vh_comm_t* c, float* buf, long count, void* stream) {
    if (!c || !buf) return cfail(VH_E_ARG, "vh_comm_allreduce: null argument");
    if (!c->connected) return cfail(VH_E_COMM, "vh_comm_allreduce: not connected");
    if (count < 0 || (size_t)count > c->cap) return cfail(VH_E_SHAPE, "vh_comm_allreduce: message above the capacity");
    if (count == 0) return VH_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    c->calls += 1;
    uint32_t tag = (uint32_t)c->calls;
    if (tag == 0) {                                        // the tag wrapped: same call on every rank (collective API)
        const int rc = comm_rewind(c, st);
        if (rc != VH_OK) return rc;
        c->calls += 1;                                     // tag 0 means "never written": skip it (keeps the parity alternating
        tag = (uint32_t)c->calls;                          // only approximately here, which is safe right after the re-zero)
    }
    const size_t par = (size_t)(c->calls & 1);
    uint64_t* local = c->local + par * c->region;
    Peers peers{};
    for (int r = 0; r < c->world; ++r) peers.p[r] = c->peer[r] + par * c->region;
    if (c->loopback && count > VH_COMM_ONESHOT_MAX) return cfail(VH_E_SHAPE, "vh_comm_allreduce: a loop-back communicator carries one-shot (decode) messages only");
    if (count <= VH_COMM_ONESHOT_MAX) {
        const size_t cap1 = c->cap < VH_COMM_ONESHOT_MAX ? c->cap : VH_COMM_ONESHOT_MAX;
        const int grid = (int)((count + 255) / 256);
        hipLaunchKernelGGL(k_ar_oneshot, dim3(grid), dim3(256), 0, st, buf, count, peers, local, cap1, c->rank, c->world,
                           tag, c->err, c->loopback ? 1 : 0);
    } else {
        BulkGeom g{};
        g.slice = (((count + c->world - 1) / c->world) + 3) & ~3L;         // multiple of 4: chunks start on 16-byte boundaries
        g.nchunk = (int)((g.slice + VH_COMM_CHUNK - 1) / VH_COMM_CHUNK);
        g.slice_pad = (long)g.nchunk * VH_COMM_CHUNK;
        g.maxchunk = c->maxchunk;
        g.offFA = c->oneshot_units;
        g.offFB = g.offFA + (size_t)c->world * g.maxchunk;
        g.offA = g.offFB + (size_t)c->world * g.maxchunk;                  // even: payload rows start on 16-byte boundaries
        g.offB = g.offA + ((size_t)c->world * g.slice_pad) / 2;            // fp32 pairs per 8-byte unit
        if (g.nchunk > g.maxchunk || g.offB + ((size_t)c->world * g.slice_pad) / 2 > c->region)
            return cfail(VH_E_SHAPE, "vh_comm_allreduce: bulk layout above the region");
        // Residency.  A block of this kernel SPINS on its peers' flags (phase 2 on every rank's phase 1, phase 3 on the
        // owners' phase 2; no wait is circular, whatever the grid), so its blocks hold CU slots until every rank has contributed.
        // One rank per device: <= 128 four-wave blocks sit next to the compute stream's GEMM (whose 12-wave blocks take a CU's whole
        // register file, one per CU: the second half's GEMM needs CUs WITHOUT a spinning block — 128 blocks leave at least half of
        // the 256 free).  Ranks SHARING a device (tests: up to 8 processes on one GPU) must leave that half free TOGETHER, or the
        // slowest rank's GEMM finds no empty CU while the other ranks' blocks spin on its contribution (r05: world 8 at 32 layers,
        // S = 552, timed out in phase 3 with 8 x 128 spinning blocks): the cap is divided by the ranks on the device.
        int cap_blocks = 128;
        if (c->ranks_per_device > 1) {
            cap_blocks = vh_num_cus() / 2 / c->ranks_per_device;
            if (cap_blocks < 1) cap_blocks = 1;
            if (cap_blocks > 128) cap_blocks = 128;
        }
        int grid = c->world * g.nchunk;
        if (grid > cap_blocks) grid = cap_blocks;
        const int vec = (reinterpret_cast<uintptr_t>(buf) & 15) == 0 ? 1 : 0;
        hipLaunchKernelGGL(k_ar_bulk, dim3(grid), dim3(256), 0, st, buf, count, peers, local, g, c->rank, c->world, tag, c->err, vec);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cfail(VH_E_HIP, "vh_comm_allreduce: launch", e);
    return VH_OK;
}

}  // extern "C"

int vh_comm_ranks_per_device(const vh_comm* c) { return c ? c->ranks_per_device : 1; }
int vh_comm_is_loopback(const vh_comm* c) { return c && c->loopback ? 1 : 0; }

// One all-reduce fused into the decode kernels: advances the call counter exactly as vh_comm_allreduce does (the two kinds
// interleave freely) and describes the exchange for the producer and the consumer launch.  `which` picks the result vector
// (0 attention sub-block, 1 MoE sub-block); `consumer_blocks` bounds the reducer count.  Not part of the public C ABI.
int vh_comm_xchg_next(vh_comm* c, long count, int which, int consumer_blocks, VhXchg* out, void* stream) {
    if (!c || !out || !c->connected) return cfail(VH_E_COMM, "vh_comm_xchg_next: not connected");
    const size_t cap1 = c->cap < VH_COMM_ONESHOT_MAX ? c->cap : VH_COMM_ONESHOT_MAX;
    if (count < 2 || (count & 1) || (size_t)count > cap1 || consumer_blocks < 1 || which < 0 || which > 1)
        return cfail(VH_E_SHAPE, "vh_comm_xchg_next: fused exchanges carry an even count within the one-shot capacity");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    c->calls += 1;
    uint32_t tag = (uint32_t)c->calls;
    if (tag == 0) {
        const int rc = comm_rewind(c, st);
        if (rc != VH_OK) return rc;
        c->calls += 1;
        tag = (uint32_t)c->calls;
    }
    const size_t par = (size_t)(c->calls & 1);
    VhXchg x{};
    for (int r = 0; r < c->world; ++r) x.peer[r] = c->peer[r] + par * c->region;
    x.local = c->local + par * c->region;
    x.reduced_g = c->reduced_g[which];
    x.err = c->err;
    x.cap = cap1;
    x.rank = c->rank; x.world = c->world; x.tag = tag;
    x.nred = consumer_blocks < 16 ? consumer_blocks : 16;
    if (x.nred > (int)(count / 2)) x.nred = (int)(count / 2);
    x.count = (int)count;
    x.loopback = c->loopback ? 1 : 0;
    *out = x;
    return VH_OK;
}

extern "C" {

int vh_comm_status(vh_comm_t* c) {
    if (!c) return -1;
    int v = 0;
    if (hipMemcpy(&v, c->err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;   // 0 = no spin ever timed out; 1 = one-shot, 2 / 3 = bulk reduce / gather phase, 4 = barrier, 5 / 6 = fused exchange
}

void vh_comm_destroy(vh_comm_t* c) {
    if (!c) return;
    for (int r = 0; r < c->world; ++r)
        if (c->opened[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    (void)hipFree(c->local);
    (void)hipFree(c->err);
    (void)hipFree(c->reduced_g[0]);
    delete c;
}

}  // extern "C"
