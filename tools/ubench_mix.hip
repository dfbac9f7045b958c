// Micro-benchmark (development tool): what the hash's permutation costs in CYCLES on gfx950, and the clock the chip
// holds under that load.  Each wave reads the shader-clock counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime) around its mix loop: cycles per mix2 per wave at 1 and at several waves per SIMD, and
// (s_memtime ticks) / (s_memrealtime ticks) = the actual shader clock in units of 100 MHz.
// Variants: the product's mix2_t (ring add = one v_add3_u32 per word); the ring add as two plain adds per word; and the S-box of NLDS of
// the 32 words of a mix looked up in a 256-byte LDS table (one dword per bank, conflict free by construction when
// lanes read different bytes of it) while the VALU does the rest -- VERDICT r02 item 8's co-issue experiment.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Istark_rs_amd/csrc tools/ubench_mix.hip -o tools/ubench_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "hash_core.h"

__device__ __forceinline__ uint32_t lds_sbox_pair(uint32_t x, const uint8_t *tab) {
    // both bytes of a word through the table: two byte addresses (mask, extract), two 8-bit LDS reads into the
    // halves of one register
    const uint32_t a0 = x & 0xFFu, a1 = (x >> 16) & 0xFFu;
    return (uint32_t)tab[a0] | ((uint32_t)tab[a1] << 16);
}
template <int VARIANT> __device__ __forceinline__ void one_mix(hashc::State2 &st, const hashc::MixK &K, const uint8_t *tab = nullptr) {
    if constexpr (VARIANT == 0) {
        hashc::mix2_t<true>(st, K);
    } else if constexpr (VARIANT >= 100) {   // NLDS = VARIANT - 100 words of the S-box through LDS
        constexpr int NLDS = VARIANT - 100;
        constexpr hashc::Consts2 C = hashc::make_consts2();
        uint32_t *s = st.s;
        uint32_t r[32];
#pragma unroll
        for (int w = 0; w < 32; w++) {
            if (w % (32 / NLDS) == 0) {
                r[w] = lds_sbox_pair(s[w] + C.rc[w], tab);     // pending round constants applied, then rotl1(251 b) from the table
            } else {
                const uint32_t t = hashc::pk_mad_u16(s[w], K.k502, C.rc502[w]);
                r[w] = hashc::bfi32(K.kFE, t, t >> 8);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t t0 = r[4 * q], t1 = r[4 * q + 1], t2 = r[4 * q + 2], t3 = r[4 * q + 3];
            const uint32_t T = hashc::xor3(hashc::xor3(t0, t1, t2), t3, K.k63);
            s[4 * q] = (T ^ t2) & K.kFF; s[4 * q + 1] = (T ^ t1) & K.kFF; s[4 * q + 2] = (T ^ t3) & K.kFF; s[4 * q + 3] = (T ^ t0) & K.kFF;
        }
        uint32_t N[32];
        N[0] = s[0] + s[1] + s[31];
#pragma unroll
        for (int w = 1; w < 31; w++) N[w] = hashc::add3(N[w - 1], s[w], s[w + 1]);
        N[31] = s[31] + N[0] + N[30];
#pragma unroll
        for (int w = 0; w < 32; w++) s[w] = N[w];
    } else {   // the same mix with the ring add spelled as two plain adds per word (asm keeps the compiler from re-fusing)
        constexpr hashc::Consts2 C = hashc::make_consts2();
        uint32_t *s = st.s;
        uint32_t r[32];
#pragma unroll
        for (int w = 0; w < 32; w++) {
            const uint32_t t = hashc::pk_mad_u16(s[w], K.k502, C.rc502[w]);
            r[w] = hashc::bfi32(K.kFE, t, t >> 8);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t t0 = r[4 * q], t1 = r[4 * q + 1], t2 = r[4 * q + 2], t3 = r[4 * q + 3];
            const uint32_t T = hashc::xor3(hashc::xor3(t0, t1, t2), t3, K.k63);
            s[4 * q] = (T ^ t2) & K.kFF; s[4 * q + 1] = (T ^ t1) & K.kFF; s[4 * q + 2] = (T ^ t3) & K.kFF; s[4 * q + 3] = (T ^ t0) & K.kFF;
        }
        uint32_t N[32];     // pair sums off the dependent chain, one plain add per word on it
        N[0] = s[0] + s[1] + s[31];
#pragma unroll
        for (int w = 1; w < 31; w++) N[w] = N[w - 1] + hashc::pair_sum(s[w], s[w + 1]);
        N[31] = s[31] + N[0] + N[30];
#pragma unroll
        for (int w = 0; w < 32; w++) s[w] = N[w];
    }
}

template <int VARIANT> __global__ __launch_bounds__(256) void k(uint32_t *out, uint64_t *clk, int mixes) {
    extern __shared__ uint32_t lds[];
    __shared__ uint8_t tab[256];
    if (VARIANT >= 100) {
        tab[threadIdx.x & 255] = (uint8_t)((((threadIdx.x & 255) * 251u) << 1) | ((((threadIdx.x & 255) * 251u) & 0xFFu) >> 7));   // rotl1(251 b)
        __syncthreads();
    }
    hashc::State2 st;
    for (int w = 0; w < 32; w++) st.s[w] = (threadIdx.x * 2654435761u + w * 40503u + blockIdx.x) & 0x00FF00FFu;
    const hashc::MixK K = hashc::mix_consts();
    const uint64_t c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < mixes; i++) one_mix<VARIANT>(st, K, tab);
    const uint64_t c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    uint32_t x = 0;
    for (int w = 0; w < 32; w++) x ^= st.s[w];
    if (x == 0x12345678u) lds[threadIdx.x] = x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) {
        const size_t wv = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        clk[2 * wv] = c1 - c0;
        clk[2 * wv + 1] = r1 - r0;
    }
}

template <int VARIANT> void run(const char *name, int blocks, int threads, int lds_kb, int mixes) {
    uint32_t *d;
    uint64_t *dc;
    const size_t waves = (size_t)blocks * (threads / 64);
    (void)hipMalloc(&d, (size_t)blocks * threads * 4);
    (void)hipMalloc(&dc, waves * 16);
    (void)hipFuncSetAttribute((const void *)k<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) k<VARIANT><<<blocks, threads, lds_kb * 1024>>>(d, dc, mixes);   // warm: clocks settle
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<VARIANT><<<blocks, threads, lds_kb * 1024>>>(d, dc, mixes);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(2 * waves);
    (void)hipMemcpy(h.data(), dc, waves * 16, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (size_t i = 0; i < waves; i++) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k<VARIANT>, threads, lds_kb * 1024);
    printf("%-22s blocks %6d x %4d  wg/CU %d  %8.3f ms  %7.2f G mix/s   %8.1f shader cycles per mix2 per wave   clock %.3f GHz\n", name, blocks,
           threads, occ, ms, 2.0 * blocks * threads * mixes / ms / 1e6, cyc / waves / mixes, cyc / real * 0.1);
    (void)hipFree(d); (void)hipFree(dc);
}

int main() {
    const int mixes = 512;
    // one wave per SIMD (4 waves per CU, 160 KB LDS per workgroup): pure issue + dependency latency of a single wave
    run<0>("mix2 1 wave/SIMD", 256, 256, 159, mixes);
    run<0>("mix2 2 waves/SIMD", 512, 256, 80, mixes);
    run<0>("mix2 4 waves/SIMD", 1024 * 4, 256, 40, mixes);
    run<0>("mix2 full", 256 * 40, 256, 0, mixes);
    run<1>("add+add 1 wave/SIMD", 256, 256, 159, mixes);
    run<1>("add+add 4 waves/SIMD", 1024 * 4, 256, 40, mixes);
    run<1>("add+add full", 256 * 40, 256, 0, mixes);
    run<104>("lds sbox 4/32 full", 256 * 40, 256, 0, mixes);
    run<108>("lds sbox 8/32 full", 256 * 40, 256, 0, mixes);
    run<116>("lds sbox 16/32 full", 256 * 40, 256, 0, mixes);
    return 0;
}
