/* ORACLE (test infrastructure): host twin of the device's counter-based synthetic-weight generator
 * (vita_amd/csrc/vh_elem.hip: hash_bf16).  Same integer arithmetic -> the fp32 oracle and the bf16 device copy hold
 * identical values at any tensor size.  Built by oracle/Makefile into oracle/_build/liboracle.so. */
#include <stdint.h>

static inline float hash_value(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const int v = (int)(z & 63) + (int)((z >> 6) & 63) + (int)((z >> 12) & 63) + (int)((z >> 18) & 63) - 126;
    return (float)v * 0.00048828125f;
}

/* dst[r * ld_dst + c] = value(seed, idx0 + r * ld_src + c) */
void hash_fill_f32(float* dst, int64_t rows, int64_t cols, int64_t ld_dst, int64_t ld_src, int64_t idx0, uint64_t seed) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        float* d = dst + r * ld_dst;
        const uint64_t base = (uint64_t)(idx0 + r * ld_src);
        for (int64_t c = 0; c < cols; ++c) d[c] = hash_value(seed, base + (uint64_t)c);
    }
}
