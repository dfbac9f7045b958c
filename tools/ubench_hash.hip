// Micro-benchmark (development tool): mix_state throughput vs occupancy on gfx950.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Istark_rs_amd/csrc tools/ubench_hash.hip -o tools/ubench_hash
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "hash_core.h"

__global__ __launch_bounds__(256) void k(uint32_t *out, int mixes) {
    extern __shared__ uint32_t lds[];
    hashc::State st;
    for (int w = 0; w < 16; w++) st.s[w] = (threadIdx.x * 2654435761u + w * 40503u + blockIdx.x) & 0x00FF00FFu;
#pragma unroll 1
    for (int i = 0; i < mixes; i++) hashc::mix_t<true>(st);
    uint32_t s = 0;
    for (int w = 0; w < 16; w++) s ^= st.s[w];
    if (s == 0x12345678u) lds[threadIdx.x] = s;   // keep the LDS allocation alive
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k2(uint32_t *out, int mixes) {
    extern __shared__ uint32_t lds[];
    hashc::State2 st;
    for (int w = 0; w < 32; w++) st.s[w] = (threadIdx.x * 2654435761u + w * 40503u + blockIdx.x) & 0x00FF00FFu;
#pragma unroll 1
    for (int i = 0; i < mixes; i++) hashc::mix2_t<true>(st);
    uint32_t s = 0;
    for (int w = 0; w < 32; w++) s ^= st.s[w];
    if (s == 0x12345678u) lds[threadIdx.x] = s;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    uint32_t *d;
    const int blocks = 256 * 40, mixes = 512;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int lds_kb[] = {0, 20, 32, 40, 53, 80, 160};
    for (int l : lds_kb) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<<<blocks, 256, l * 1024>>>(d, mixes);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<<<blocks, 256, l * 1024>>>(d, mixes);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double n_mix = (double)blocks * 256 * mixes;
        int occ = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, l * 1024);
        printf("lds %3d KB  blocks/CU %d  %8.3f ms  %6.2f G mix/s  (%.1f ns per mix per CU-wave-slot)\n", l, occ, ms, n_mix / ms / 1e6, 0.0);
    }
    for (int l : lds_kb) {   // two hashes per state: count both
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k2<<<blocks / 2, 256, l * 1024>>>(d, mixes);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k2<<<blocks / 2, 256, l * 1024>>>(d, mixes);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        int occ = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k2, 256, l * 1024);
        printf("pairs: lds %3d KB  blocks/CU %d  %8.3f ms  %6.2f G mix/s\n", l, occ, ms, (double)blocks * 256 * mixes / ms / 1e6);
    }
    return 0;
}
