// Micro-benchmark (development tool): HBM throughput of the NTT passes' access patterns without
// any arithmetic.  A workgroup moves a tile of R rows x W elements; rows are B elements apart.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_stride.hip -o tools/ubench_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int LOGR, int LOGW, int NT_LD = 0, int NT_ST = 0> __global__ __launch_bounds__(1 << (LOGR + LOGW - 4)) void strided(const uint32_t *in, uint32_t *out, uint32_t blog, uint32_t n_tiles) {
    constexpr int NT = 1 << (LOGR + LOGW - 4), W = 1 << LOGW;
    const uint32_t tid = threadIdx.x, w = tid & (W - 1), j0 = tid >> LOGW;
    const uint32_t b = blockIdx.x;
    const uint32_t tix = (n_tiles % 8u) ? b : (b & 7u) * (n_tiles >> 3) + (b >> 3);
    const uint32_t tpa = 1u << (blog - LOGW);
    const size_t base = ((size_t)(tix / tpa) << (blog + LOGR)) + ((size_t)(tix % tpa) << LOGW);
    uint32_t v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t *p = in + base + ((size_t)(j0 + i * (NT >> LOGW)) << blog) + w;
        v[i] = NT_LD ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint32_t *p = out + base + ((size_t)(j0 + i * (NT >> LOGW)) << blog) + w;
        if (NT_ST) __builtin_nontemporal_store(v[i] + 1, p); else *p = v[i] + 1;
    }
}
template <int U, int NT_LD, int NT_ST> __global__ __launch_bounds__(512) void linear(const uint4 *in, uint4 *out, size_t n4) {
    size_t i = (size_t)blockIdx.x * 512 * U + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 512 * U;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    for (; i < n4; i += step) {
        u4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT_LD ? __builtin_nontemporal_load((const u4 *)in + i + k * 512) : ((const u4 *)in)[i + k * 512];
#pragma unroll
        for (int k = 0; k < U; k++) { v[k].x++; if (NT_ST) __builtin_nontemporal_store(v[k], (u4 *)out + i + k * 512); else ((u4 *)out)[i + k * 512] = v[k]; }
    }
}
template <class F> void timeit(const char *name, size_t bytes, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; r++) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.1f us  %7.1f GB/s (read+write)\n", name, 100.0 * ms, 2.0 * bytes * 10 / ms / 1e6);
}
int main() {
    const uint32_t L = 27;   // 2^27 elements = 512 MiB per buffer (4 columns of 2^25)
    const size_t n = (size_t)1 << L;
    uint32_t *a, *b;
    (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4);
    (void)hipMemset(a, 1, n * 4);
    timeit("linear uint4 copy U=1, 2048 blocks", n * 4, [&] { linear<1, 0, 0><<<256 * 8, 512>>>((const uint4 *)a, (uint4 *)b, n / 4); });
    timeit("linear uint4 copy U=4, 1024 blocks", n * 4, [&] { linear<4, 0, 0><<<256 * 4, 512>>>((const uint4 *)a, (uint4 *)b, n / 4); });
    timeit("linear uint4 copy U=4, one pass", n * 4, [&] { linear<4, 0, 0><<<(unsigned)(n / 4 / 2048), 512>>>((const uint4 *)a, (uint4 *)b, n / 4); });
    timeit("linear U=4 one pass, nt loads", n * 4, [&] { linear<4, 1, 0><<<(unsigned)(n / 4 / 2048), 512>>>((const uint4 *)a, (uint4 *)b, n / 4); });
    timeit("linear U=4 one pass, nt stores", n * 4, [&] { linear<4, 0, 1><<<(unsigned)(n / 4 / 2048), 512>>>((const uint4 *)a, (uint4 *)b, n / 4); });
    timeit("linear U=4 one pass, nt both", n * 4, [&] { linear<4, 1, 1><<<(unsigned)(n / 4 / 2048), 512>>>((const uint4 *)a, (uint4 *)b, n / 4); });
    { const uint32_t tiles = (uint32_t)(n >> 13);
      timeit("strided 8,5 B=2^16 nt loads", n * 4, [&] { strided<8, 5, 1, 0><<<tiles, 512>>>(a, b, 16, tiles); });
      timeit("strided 8,5 B=2^16 nt stores", n * 4, [&] { strided<8, 5, 0, 1><<<tiles, 512>>>(a, b, 16, tiles); });
      timeit("strided 8,5 B=2^16 nt both", n * 4, [&] { strided<8, 5, 1, 1><<<tiles, 512>>>(a, b, 16, tiles); }); }
#define S(LR, LW, BLOG) { char nm[64]; snprintf(nm, sizeof nm, "strided R=2^%d W=%d (%d B runs) B=2^%d", LR, 1 << LW, 4 << LW, BLOG); \
        const uint32_t tiles = (uint32_t)(n >> (LR + LW)); \
        timeit(nm, n * 4, [&] { strided<LR, LW><<<tiles, 1 << (LR + LW - 4)>>>(a, b, BLOG, tiles); }); }
    S(8, 5, 8) S(8, 5, 16) S(8, 5, 17) S(9, 5, 16) S(8, 4, 16) S(8, 6, 16) S(7, 6, 16) S(7, 7, 16) S(9, 3, 16) S(6, 6, 16) S(6, 7, 16) S(6, 8, 16)
    return 0;
}
