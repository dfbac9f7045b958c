// Development tool: hashq::node_hash (one hash over a quad of lanes) against hashc::node_hash on the
// device, and the latency of a chain of dependent node hashes in a single wave either way.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Istark_rs_amd/csrc tools/quad_hash_test.hip -o tools/quad_hash_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "hash_quad.h"

__global__ void single_kernel(const uint32_t *in, uint32_t *out, int n, int chain) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t l[8], r[8], d[8];
    for (int j = 0; j < 8; j++) { l[j] = in[16 * i + j]; r[j] = in[16 * i + 8 + j]; }
    for (int c = 0; c < chain; c++) {
        hashc::node_hash<8>(l, r, d);
        for (int j = 0; j < 8; j++) l[j] = d[j];
    }
    for (int j = 0; j < 8; j++) out[8 * i + j] = d[j];
}
__global__ void quad_kernel(const uint32_t *in, uint32_t *out, int n, int chain) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 2;
    if (i >= n) return;
    const hashq::Lane L = hashq::make_lane(threadIdx.x);
    uint32_t l[8], r[8], lo = 0, hi = 0;
    for (int j = 0; j < 8; j++) { l[j] = in[16 * i + j]; r[j] = in[16 * i + 8 + j]; }
    for (int c = 0; c < chain; c++) {
        hashq::node_hash(l, r, L, lo, hi);
        l[0] = hashq::bcast<0>(lo); l[1] = hashq::bcast<1>(lo); l[2] = hashq::bcast<2>(lo); l[3] = hashq::bcast<3>(lo);
        l[4] = hashq::bcast<0>(hi); l[5] = hashq::bcast<1>(hi); l[6] = hashq::bcast<2>(hi); l[7] = hashq::bcast<3>(hi);
    }
    out[8 * i + L.q] = lo;
    out[8 * i + 4 + L.q] = hi;
}

int main() {
    const int n = 1000;
    std::vector<uint32_t> h(16 * n);
    uint64_t z = 12345;
    for (auto &x : h) { z = z * 6364136223846793005ull + 1442695040888963407ull; x = (uint32_t)(z >> 32); }
    for (int j = 0; j < 16; j++) h[j] = 0;                 // all-zero children
    for (int j = 0; j < 16; j++) h[16 + j] = 0xFFFFFFFFu;  // all-ones children
    uint32_t *din, *da, *db;
    (void)hipMalloc(&din, h.size() * 4); (void)hipMalloc(&da, 8 * n * 4); (void)hipMalloc(&db, 8 * n * 4);
    (void)hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int chain : {1, 3}) {
        single_kernel<<<(n + 63) / 64, 64>>>(din, da, n, chain);
        quad_kernel<<<(4 * n + 255) / 256, 256>>>(din, db, n, chain);
        std::vector<uint32_t> a(8 * n), b(8 * n);
        (void)hipMemcpy(a.data(), da, a.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(b.data(), db, b.size() * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 8 * n; i++) bad += a[i] != b[i];
        printf("chain %d: %d of %d digest words differ (first digest %08x %08x vs %08x %08x)\n", chain, bad, 8 * n, a[0], a[1], b[0], b[1]);
    }
    // latency: one wave, a chain of dependent node hashes
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int chain = 200;
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            if (which == 0) single_kernel<<<1, 64>>>(din, da, 64, chain);
            else quad_kernel<<<1, 64>>>(din, db, 16, chain);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
        }
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f us per node hash in a single wave\n", which ? "quad lanes " : "single lane", 1e3 * ms / chain);
    }
    return bad != 0;
}
