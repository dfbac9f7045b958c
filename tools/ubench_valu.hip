// Micro-benchmark (development tool): VALU issue rates on gfx950, one instruction kind per kernel.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
// 8 independent dependency chains per thread, 8 waves per SIMD: measures issue rate, not latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 8192
#define ASM1(op) asm volatile(op " %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(b)); x[i] = t;
#define ASM3(op) asm volatile(op " %0, %1, %2, %3" : "=v"(t) : "v"(x[i]), "v"(b), "v"(c)); x[i] = t;
template <int OP> __global__ void k(uint32_t *out, uint32_t a0, uint32_t b0) {
    uint32_t x[8];
    for (int i = 0; i < 8; i++) x[i] = a0 + threadIdx.x * 7 + i;
    uint32_t b = b0 | 1, c = b0 * 77 + 5;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t t;
            if (OP == 0) { ASM1("v_add_u32") }
            if (OP == 1) { ASM1("v_xor_b32") }
            if (OP == 2) { ASM1("v_mul_lo_u32") }
            if (OP == 3) { ASM1("v_mul_hi_u32") }
            if (OP == 4) { ASM3("v_add3_u32") }
            if (OP == 5) { ASM3("v_bfi_b32") }
            if (OP == 6) { asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(t) : "v"(x[i]), "v"(b), "v"(c)); x[i] = t; }
            if (OP == 7) { ASM3("v_perm_b32") }
            if (OP == 8) { ASM3("v_alignbit_b32") }
            if (OP == 9) { ASM3("v_lshl_add_u32") }
            if (OP == 10) { ASM3("v_pk_mad_u16") }
            if (OP == 11) { ASM1("v_pk_add_u16") }
            if (OP == 12) { ASM1("v_pk_mul_lo_u16") }
            if (OP == 13) { ASM3("v_fma_f32") }
            if (OP == 14) { ASM1("v_add_f32") }
            if (OP == 15) { ASM1("v_min_u32") }
            if (OP == 16) { ASM1("v_lshrrev_b32") }
            if (OP == 17) { ASM3("v_mad_u32_u24") }
            if (OP == 18) { ASM1("v_mul_u32_u24") }
            if (OP == 19) { ASM3("v_and_or_b32") }
            if (OP == 20) { ASM1("v_sub_u32") }
            if (OP == 21) { ASM3("v_xad_u32") }
            if (OP == 22) { ASM1("v_pk_lshrrev_b16") }
            if (OP == 23) { ASM3("v_mad_i32_i24") }
            if (OP == 24) { ASM1("v_and_b32") }
            if (OP == 25) { ASM1("v_or_b32") }
            if (OP == 26) { ASM1("v_lshlrev_b32") }
            if (OP == 27) { ASM1("v_ashrrev_i32") }
            if (OP == 28) { asm volatile("v_bfe_u32 %0, %1, 3, 8" : "=v"(t) : "v"(x[i])); x[i] = t; }
            if (OP == 29) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(t) : "v"(x[i]), "v"(b)); x[i] = t; }
            if (OP == 30) { asm volatile("v_not_b32 %0, %1" : "=v"(t) : "v"(x[i])); x[i] = t; }
            if (OP == 31) { asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(x[i])); x[i] = t ^ 1; }
            if (OP == 32) { ASM1("v_max_u32") }
            if (OP == 33) { ASM1("v_min_i32") }
            if (OP == 34) { ASM3("v_med3_u32") }
            if (OP == 35) { ASM3("v_sad_u32") }
            if (OP == 36) { ASM3("v_dot4_u32_u8") }
            if (OP == 37) { ASM3("v_lshl_or_b32") }
            if (OP == 38) { ASM3("v_add_lshl_u32") }
            if (OP == 39) { ASM3("v_or3_b32") }
            if (OP == 40) { ASM1("v_xnor_b32") }
            if (OP == 41) { asm volatile("v_add_co_u32 %0, vcc, %1, %2" : "=v"(t) : "v"(x[i]), "v"(b) : "vcc"); x[i] = t; }
            if (OP == 42) { asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(t) : "v"(x[i]), "v"(b) : "vcc"); x[i] = t; }
            if (OP == 43) { ASM1("v_sub_f32") }
            if (OP == 44) { ASM1("v_mul_f32") }
            if (OP == 45) { asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(t) : "v"(x[i])); x[i] = t; }
            if (OP == 46) { asm volatile("v_add_u32 %0, %1, %2 row_shr:1" : "=v"(t) : "v"(x[i]), "v"(b)); x[i] = t; }
            if (OP == 47) { asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(t) : "v"(x[i]), "v"(b)); x[i] = t; }
            if (OP == 48) { ASM3("v_mad_u16") }
            if (OP == 49) { ASM1("v_mul_lo_u16") }
            if (OP == 50) { ASM1("v_add_u16") }
            // operand variety: all sources are different, changing registers
            if (OP == 60) { asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(t) : "v"(x[i]), "v"(x[(i + 1) & 7]), "v"(x[(i + 3) & 7])); x[i] = t; }
            if (OP == 61) { asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(x[(i + 1) & 7])); x[i] = t; }
            if (OP == 62) { asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(x[(i + 5) & 7])); x[i] = t; }
            if (OP == 63) { asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(t) : "v"(x[i]), "v"(x[(i + 1) & 7]), "v"(x[(i + 3) & 7])); x[i] = t; }
            // mixes: one fast + one slow instruction per step (counted as 2)
            if (OP == 64) { asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(b)); asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x[i]) : "v"(t), "v"(c)); }
            if (OP == 65) { asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(b)); asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(t), "v"(c), "v"(b)); }
            if (OP == 66) { asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(b)); asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[i]) : "v"(t), "v"(c)); }
            // scalar operand / inline constant forms
            if (OP == 67) { asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "s"(b0), "v"(x[i])); x[i] = t; }
            if (OP == 68) { asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(t) : "v"(x[i])); x[i] = t + 1; }
            if (OP == 69) { asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(t) : "v"(x[i])); x[i] = t + 1; }
            if (OP == 70) { asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(t) : "s"(b0), "v"(x[i]), "v"(x[(i + 3) & 7])); x[i] = t; }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 64-bit-per-lane packed ops
template <int OP> __global__ void k2(uint32_t *out, uint32_t a0, uint32_t b0) {
    uint64_t x[4];
    for (int i = 0; i < 4; i++) x[i] = ((uint64_t)(a0 + threadIdx.x) << 32) | (a0 + i);
    uint64_t b = ((uint64_t)b0 << 32) | (b0 + 3), c = b * 3;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint64_t t;
            if (OP == 0) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(x[i]), "v"(b), "v"(c)); x[i] = t; }
            if (OP == 1) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(b)); x[i] = t; }
            if (OP == 2) { asm volatile("v_lshlrev_b64 %0, 3, %1" : "=v"(t) : "v"(x[i])); x[i] = t; }
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 4; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

template <class K> void run(const char *name, K kern, double ops_per_thread_iter) {
    uint32_t *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 blocks of 256 per CU = 8 waves per SIMD
    kern<<<blocks, 256>>>(d, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(d, 3, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)blocks * 256 * ITERS * ops_per_thread_iter;
    printf("%-20s %8.3f ms  %7.2f T lane-instr/s\n", name, ms, lane_ops / ms / 1e9);
    hipFree(d);
}
#define R(OP, NAME) run(NAME, k<OP>, 8)
int main() {
    R(0, "v_add_u32"); R(20, "v_sub_u32"); R(1, "v_xor_b32"); R(15, "v_min_u32"); R(16, "v_lshrrev_b32");
    R(2, "v_mul_lo_u32"); R(3, "v_mul_hi_u32"); R(18, "v_mul_u32_u24"); R(17, "v_mad_u32_u24"); R(23, "v_mad_i32_i24");
    R(4, "v_add3_u32"); R(9, "v_lshl_add_u32"); R(21, "v_xad_u32"); R(19, "v_and_or_b32");
    R(5, "v_bfi_b32"); R(6, "v_bitop3_b32"); R(7, "v_perm_b32"); R(8, "v_alignbit_b32");
    R(10, "v_pk_mad_u16"); R(11, "v_pk_add_u16"); R(12, "v_pk_mul_lo_u16"); R(22, "v_pk_lshrrev_b16");
    R(13, "v_fma_f32"); R(14, "v_add_f32"); R(43, "v_sub_f32"); R(44, "v_mul_f32"); R(45, "v_cvt_f32_u32");
    R(24, "v_and_b32"); R(25, "v_or_b32"); R(40, "v_xnor_b32"); R(30, "v_not_b32"); R(31, "v_mov_b32(+xor)");
    R(26, "v_lshlrev_b32"); R(27, "v_ashrrev_i32"); R(28, "v_bfe_u32"); R(29, "v_cndmask_b32"); R(42, "v_cmp+v_cndmask (2)");
    R(32, "v_max_u32"); R(33, "v_min_i32"); R(34, "v_med3_u32"); R(35, "v_sad_u32"); R(36, "v_dot4_u32_u8");
    R(37, "v_lshl_or_b32"); R(38, "v_add_lshl_u32"); R(39, "v_or3_b32"); R(41, "v_add_co_u32");
    R(46, "v_add_u32 dpp"); R(47, "v_add_u32 sdwa"); R(48, "v_mad_u16"); R(49, "v_mul_lo_u16"); R(50, "v_add_u16");
    R(60, "bitop3 3 distinct"); R(61, "add 2 distinct"); R(62, "xor 2 distinct"); R(63, "add3 3 distinct");
    run("add+mul_lo (2)", k<64>, 16); run("xor+add3 (2)", k<65>, 16); run("xor+add (2)", k<66>, 16);
    R(67, "v_add_u32 sgpr"); run("lshl imm (+add)", k<68>, 16); run("lshr imm (+add)", k<69>, 16); R(70, "bitop3 sgpr,v,v");
    run("v_pk_fma_f32", k2<0>, 4); run("v_pk_add_f32", k2<1>, 4); run("v_lshlrev_b64", k2<2>, 4);
    return 0;
}
