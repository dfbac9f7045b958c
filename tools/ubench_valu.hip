// Micro-benchmark (development tool): VALU integer-multiply issue rates on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 4096
template <int OP> __global__ void k(uint32_t *out, uint32_t a0, uint32_t b0) {
    uint32_t x[8];
    for (int i = 0; i < 8; i++) x[i] = a0 + threadIdx.x * 7 + i;
    uint32_t b = b0 | 1;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) x[i] = x[i] * b + 1;                                             // v_mul_lo_u32 (+add)
            if (OP == 1) x[i] = __umulhi(x[i], b) + 1;                                    // v_mul_hi_u32
            if (OP == 2) { uint64_t t = (uint64_t)x[i] * b + x[i]; x[i] = (uint32_t)(t >> 32) ^ (uint32_t)t; }  // v_mad_u64_u32
            if (OP == 3) x[i] = __umul24(x[i], b) + 1;                                    // v_mul_u32_u24
            if (OP == 4) x[i] = x[i] + b;                                                 // v_add_u32
            if (OP == 5) x[i] = (x[i] ^ b) + (x[i] >> 3);                                 // 3 simple ops
            if (OP == 6) { double d = (double)x[i]; d = fma(d, 1.0000001, 3.0); x[i] = (uint32_t)d; }       // f64 fma + cvts
            if (OP == 7) x[i] = __builtin_amdgcn_perm(x[i], b, 0x01020300u) + 1;          // v_perm_b32
            if (OP == 8) { float f = __uint_as_float(x[i]); f = fmaf(f, 1.0000001f, 3.0f); f = fmaf(f, 0.9999f, 1.0f); x[i] = __float_as_uint(f); }  // 2 x v_fma_f32
            if (OP == 9) { x[i] = (x[i] ^ b) & (x[i] >> 1); x[i] = (x[i] | b) ^ (x[i] << 2); }   // logic/shift mix (~6 ops)
            if (OP == 10) { uint32_t t; asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(t) : "v"(x[i]), "v"(b), "v"(x[(i + 1) & 7])); x[i] = t; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(b)); x[i] = t; }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> void run(const char *name, double ops_per_iter) {
    uint32_t *d;
    hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 blocks of 256 per CU
    k<OP><<<blocks, 256>>>(d, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 3, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double lane_ops = (double)blocks * 256 * ITERS * 8 * ops_per_iter;
    printf("%-28s %8.3f ms  %8.2f Tlane-ops/s (counting %g op/iter)\n", name, ms, lane_ops / ms / 1e9, ops_per_iter);
    hipFree(d);
}
int main() {
    run<4>("v_add_u32", 1);
    run<5>("xor+shift+add", 3);
    run<0>("v_mul_lo_u32 (+add)", 1);
    run<1>("v_mul_hi_u32 (+add)", 1);
    run<2>("v_mad_u64_u32 (+xor)", 1);
    run<3>("v_mul_u32_u24 (+add)", 1);
    run<6>("cvt+f64 fma+cvt", 1);
    run<7>("v_perm_b32 (+add)", 1);
    run<8>("2x v_fma_f32", 2);
    run<9>("logic/shift mix", 6);
    run<10>("add3 + xor (asm)", 2);
    return 0;
}
