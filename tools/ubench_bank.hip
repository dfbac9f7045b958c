// Micro-benchmark (development tool): does VGPR bank placement of the sources decide whether a
// simple VALU op issues in 2 or 4 cycles on gfx950?  Explicit physical registers.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_bank.hip -o tools/ubench_bank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
// 8 independent instructions per step; destination = first source register (in place)
#define STEP8(OP, A, B, C, D, E, F, G, H) asm volatile(OP(A) OP(B) OP(C) OP(D) OP(E) OP(F) OP(G) OP(H) ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
template <int V> __global__ void k(uint32_t *out) {
    asm volatile("v_mov_b32 v40, 1\n v_mov_b32 v41, 2\n v_mov_b32 v42, 3\n v_mov_b32 v43, 4\n v_mov_b32 v44, 5\n v_mov_b32 v45, 6\n v_mov_b32 v46, 7\n v_mov_b32 v47, 8\n"
                 "v_mov_b32 v48, 9\n v_mov_b32 v49, 10\n v_mov_b32 v50, 11\n v_mov_b32 v51, 12\n v_mov_b32 v52, 13\n v_mov_b32 v53, 14\n v_mov_b32 v54, 15\n v_mov_b32 v55, 16\n"
                 "v_mov_b32 v56, 17\n v_mov_b32 v57, 18\n v_mov_b32 v58, 19\n v_mov_b32 v59, 20\n v_mov_b32 v60, 21\n v_mov_b32 v61, 22\n v_mov_b32 v62, 23\n v_mov_b32 v63, 24\n" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
    for (int it = 0; it < ITERS; it++) {
        if (V == 0) {  // xor: sources in different banks (dst d, src d and d+1)
#define X0(d) "v_xor_b32 v" #d ", v" #d ", v57\n"
            STEP8(X0, 40, 41, 42, 43, 44, 45, 46, 47)   // v57 bank 1: conflicts only with 41, 45
        }
        if (V == 1) {  // xor: both sources in the same bank
#define X1(d) "v_xor_b32 v" #d ", v" #d ", v56\n"
            asm volatile(X1(40) X1(44) X1(48) X1(52) X1(40) X1(44) X1(48) X1(52) ::: "v40","v44","v48","v52");
        }
        if (V == 2) {  // bitop3, three banks
#define B0(d) "v_bitop3_b32 v" #d ", v" #d ", v57, v58 bitop3:0x96\n"
            asm volatile(B0(40) B0(44) B0(48) B0(52) B0(60) B0(63) B0(40) B0(44) ::: "v40","v44","v48","v52","v60","v63");
        }
        if (V == 3) {  // bitop3, all three sources in bank 0
#define B1(d) "v_bitop3_b32 v" #d ", v" #d ", v56, v60 bitop3:0x96\n"
            asm volatile(B1(40) B1(44) B1(48) B1(52) B1(40) B1(44) B1(48) B1(52) ::: "v40","v44","v48","v52");
        }
        if (V == 4) {  // bitop3, two sources share a bank
#define B2(d) "v_bitop3_b32 v" #d ", v" #d ", v56, v57 bitop3:0x96\n"
            asm volatile(B2(40) B2(44) B2(48) B2(52) B2(40) B2(44) B2(48) B2(52) ::: "v40","v44","v48","v52");
        }
        if (V == 5) {  // dependent chain: each instruction uses the previous result
            asm volatile("v_xor_b32 v40, v40, v57\n v_xor_b32 v41, v40, v58\n v_xor_b32 v42, v41, v59\n v_xor_b32 v43, v42, v56\n"
                         "v_xor_b32 v44, v43, v57\n v_xor_b32 v45, v44, v58\n v_xor_b32 v46, v45, v59\n v_xor_b32 v40, v46, v56\n" ::: "v40","v41","v42","v43","v44","v45","v46");
        }
        if (V == 6) {  // independent but dst of one = src of the one two later
            asm volatile("v_xor_b32 v40, v41, v57\n v_xor_b32 v42, v43, v58\n v_xor_b32 v41, v40, v59\n v_xor_b32 v43, v42, v56\n"
                         "v_xor_b32 v40, v41, v57\n v_xor_b32 v42, v43, v58\n v_xor_b32 v41, v40, v59\n v_xor_b32 v43, v42, v56\n" ::: "v40","v41","v42","v43");
        }
    }
    uint32_t r;
    asm volatile("v_xor_b32 %0, v40, v44\n v_xor_b32 %0, %0, v48\n v_xor_b32 %0, %0, v52" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class K> void run(const char *name, K kern) {
    uint32_t *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<256 * 8, 256>>>(d);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<256 * 8, 256>>>(d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %7.2f T lane-instr/s\n", name, ms, (double)256 * 8 * 256 * ITERS * 8 / ms / 1e9);
    (void)hipFree(d);
}
int main() {
    run("xor, sources in different banks", k<0>);
    run("xor, both sources in one bank", k<1>);
    run("bitop3, three banks", k<2>);
    run("bitop3, three sources in one bank", k<3>);
    run("bitop3, two sources share a bank", k<4>);
    run("xor, fully dependent chain", k<5>);
    run("xor, dependent at distance 2", k<6>);
    return 0;
}
