// Probe: does a counted s_waitcnt vmcnt(N) order an LDS-DMA (global_load_lds) against YOUNGER ordinary VGPR loads?
// One wave per block issues: [DMA of a cold (HBM-miss) 1 KB line set -> LDS] then [K hot (L2-hit) global_load_dwordx4 to
// VGPRs], waits vmcnt(K) — which claims "the DMA has landed" only if the two kinds retire in issue order — and reads the
// LDS bytes back.  The LDS region was pre-filled with a sentinel; a sentinel read back = the wait passed before the DMA
// landed = the kinds are NOT mutually ordered.  The other 7 waves of the block hammer LDS with ds_reads (as the GEMM's
// consumers do) to delay the DMA's LDS write.  Also the reverse direction (cold VGPR load, hot DMAs, vmcnt(K)).
//   hipcc --offload-arch=gfx950 -O3 vmcnt_order.hip -o vmcnt_order && ./vmcnt_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void glds16(const unsigned char* base, uint32_t off, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}

template <int K, int MODE>   // MODE 0: cold DMA then K hot VGPR loads; MODE 1: cold VGPR load then K hot DMAs
__global__ __launch_bounds__(512) void probe(const unsigned char* cold, size_t cold_bytes, const unsigned char* hot, int iters,
                                              unsigned long long* bad, unsigned long long* checked, uint32_t* sink) {
    __shared__ __attribute__((aligned(1024))) uint32_t lds[40960];   // 160 KB
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t)lds;
    unsigned long long nbad = 0, nchk = 0;
    uint32_t acc = 0;
    uint64_t rng = 0x9E3779B97F4A7C15ull * (blockIdx.x + 1);
    for (int it = 0; it < iters; ++it) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        const size_t line = (size_t)((rng >> 20) % (cold_bytes / 1024)) * 1024;      // block-uniform cold 1 KB chunk
        if (wid == 0) {
            lds[lane * 4 + 0] = 0xDEADBEEFu; lds[lane * 4 + 1] = 0xDEADBEEFu; lds[lane * 4 + 2] = 0xDEADBEEFu; lds[lane * 4 + 3] = 0xDEADBEEFu;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (wid == 0) {
            const unsigned char* cb = cold + line;
            u32x4 h[K];
            u32x4 cv = {0, 0, 0, 0};
            if (MODE == 0) {
                glds16(cb, lane * 16, lds0);                                   // cold DMA (oldest)
#pragma unroll
                for (int q = 0; q < K; ++q)
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(h[q]) : "v"((uint32_t)(lane * 16 + q * 1024)), "s"(hot) : "memory");
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(K) : "memory");       // "the DMA has landed" iff in-order
                const uint32_t got = lds[lane * 4];
                const uint32_t exp = (uint32_t)((line + lane * 16) / 4 * 2654435761u);
                nchk++;
                if (got != exp) nbad++;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < K; ++q) { asm volatile("" : "+v"(h[q])); acc += h[q].x; }
            } else {
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(cv) : "v"((uint32_t)(lane * 16)), "s"(cb) : "memory");   // cold VGPR load (oldest)
#pragma unroll
                for (int q = 0; q < K; ++q) glds16(hot, lane * 16 + q * 1024, lds0 + 1024 + q * 1024);
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(cv) : "i"(K) : "memory");   // "the cold load has landed" iff in-order
                const uint32_t exp = (uint32_t)((line + lane * 16) / 4 * 2654435761u);
                nchk++;
                if (cv.x != exp) nbad++;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            // LDS read pressure from the other 7 waves
#pragma unroll 8
            for (int r = 0; r < 64; ++r) acc += lds[2048 + ((lane * 4 + r * 256 + wid * 64) & 32767)];
        }
        __syncthreads();
    }
    if (wid == 0) { atomicAdd(bad, nbad); atomicAdd(checked, nchk); }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void fill(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)(i * 2654435761u);
}

int main() {
    const size_t cold_bytes = 8ull << 30, hot_bytes = 64 << 10;
    unsigned char *cold, *hot; unsigned long long *cnt; uint32_t* sink;
    hipMalloc(&cold, cold_bytes); hipMalloc(&hot, hot_bytes); hipMalloc(&cnt, 16); hipMalloc(&sink, 4);
    fill<<<4096, 256>>>((uint32_t*)cold, cold_bytes / 4); fill<<<64, 256>>>((uint32_t*)hot, hot_bytes / 4);
    hipDeviceSynchronize();
    auto run = [&](const char* name, auto kern) {
        hipMemset(cnt, 0, 16);
        kern<<<1024, 512>>>(cold, cold_bytes, hot, 400, cnt, cnt + 1, sink);
        hipError_t e = hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
        printf("%-44s checked %llu  violations %llu  (%s)\n", name, h[1], h[0], hipGetErrorString(e));
    };
    run("cold DMA, 4 hot VGPR loads, vmcnt(4)", probe<4, 0>);
    run("cold DMA, 12 hot VGPR loads, vmcnt(12)", probe<12, 0>);
    run("cold VGPR load, 4 hot DMAs, vmcnt(4)", probe<4, 1>);
    run("cold VGPR load, 12 hot DMAs, vmcnt(12)", probe<12, 1>);
    return 0;
}
