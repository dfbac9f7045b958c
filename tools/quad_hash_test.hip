// Development tool: hashq::node_hash (one hash over a quad of lanes) and hashx::node_hash (over a row of sixteen) against
// hashc::node_hash on the device, the latency of a chain of dependent node hashes in a single wave each way, and the time
// of one narrow tree level in a 1024-lane workgroup (LDS in, hash, LDS + global out, barrier) with 1 .. 64 nodes.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Istark_rs_amd/csrc tools/quad_hash_test.hip -o tools/quad_hash_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "hash_quad.h"
#include "hash_hex.h"

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

__global__ void hex_kernel(const uint32_t *in, uint32_t *out, int n, int chain) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 4;
    if (i >= n) return;
    const hashx::Lane L = hashx::make_lane(threadIdx.x);
    const uint32_t w = threadIdx.x & 15u, j = w >> 2;
    uint32_t ml = hashx::message(in[16 * i + j], in[16 * i + 4 + j], L);
    const uint32_t mr = hashx::message(in[16 * i + 8 + j], in[16 * i + 12 + j], L);
    uint32_t x = 0;
    for (int c = 0; c < chain; c++) {
        x = hashx::node_hash(ml, mr, L);
        ml = x & 0x00FF00FFu;
    }
    uint8_t *o = (uint8_t *)(out + 8 * i);
    o[w] = (uint8_t)x;
    o[16 + w] = (uint8_t)(x >> 16);
}
// one tree level of `half` nodes, `levels` times over (every round hashes the same children): MODE 0 quad, 1 hex
#define LV_MAX 2048
template <int MODE> __global__ __launch_bounds__(1024) void level_kernel(const uint32_t *in, uint4 *nodes, uint32_t half, int levels) {
    __shared__ uint32_t buf[8 * LV_MAX];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 2 * half; i += 1024)
        for (int w = 0; w < 8; w++) buf[w * LV_MAX + i] = in[8 * i + w];
    __syncthreads();
    const hashq::Lane lane = hashq::make_lane(tid);
    const hashx::Lane row = hashx::make_lane(tid);
    uint32_t base = 0;
    for (int lv = 0; lv < levels; lv++) {
        if (MODE == 0) {
            const uint32_t node = tid >> 2;
            if (node < half) {
                uint32_t l[8], r[8], lo, hi;
                for (int w = 0; w < 8; w++) { l[w] = buf[w * LV_MAX + base + 2 * node]; r[w] = buf[w * LV_MAX + base + 2 * node + 1]; }
                hashq::node_hash(l, r, lane, lo, hi);
                uint32_t *dst = (uint32_t *)(nodes + 2 * ((size_t)lv * half + node));
                dst[lane.q] = lo; dst[4 + lane.q] = hi;
                buf[lane.q * LV_MAX + (base ^ (LV_MAX / 2)) + node] = lo;
                buf[(4 + lane.q) * LV_MAX + (base ^ (LV_MAX / 2)) + node] = hi;
            }
        } else {
            const uint32_t node = tid >> 4, w = tid & 15u, j = w >> 2;
            if (node < half) {
                const uint32_t *src = buf + base + 2 * node;
                const uint32_t ml = hashx::message(src[j * LV_MAX], src[(4 + j) * LV_MAX], row);
                const uint32_t mr = hashx::message(src[j * LV_MAX + 1], src[(4 + j) * LV_MAX + 1], row);
                const uint32_t x = hashx::node_hash(ml, mr, row);
                uint8_t *dst = (uint8_t *)(nodes + 2 * ((size_t)lv * half + node));
                dst[w] = (uint8_t)x; dst[16 + w] = (uint8_t)(x >> 16);
                uint8_t *nb = (uint8_t *)(buf + (base ^ (LV_MAX / 2)) + node);
                nb[4 * (j * LV_MAX) + (w & 3u)] = (uint8_t)x;
                nb[4 * ((4 + j) * LV_MAX) + (w & 3u)] = (uint8_t)(x >> 16);
            }
        }
        __syncthreads();
        // the next round reads the same children again: copy nothing, keep base (the written slot is scratch)
    }
}

// the last launch of a tree as merkle_top_kernel<false> runs it (256 chunk roots -> root, then the Fiat-Shamir round by
// lane 0), with a time stamp (s_memrealtime, 100 MHz) after every phase: where do its ~18 us go?
__global__ __launch_bounds__(1024) void final_kernel(const uint32_t *in, uint4 *nodes, uint32_t chunk, uint32_t *fs_words, uint64_t *alpha, uint64_t *stamps) {
    __shared__ uint32_t buf[8 * LV_MAX];
    const uint32_t tid = threadIdx.x;
    int ns = 0;
    auto stamp = [&]() { if (tid == 0) stamps[ns] = __builtin_amdgcn_s_memrealtime(); ns++; };
    stamp();
    for (uint32_t i = tid; i < chunk; i += 1024) {
        const uint4 a = ((const uint4 *)in)[2 * i], b = ((const uint4 *)in)[2 * i + 1];
        buf[0 * LV_MAX + i] = a.x; buf[1 * LV_MAX + i] = a.y; buf[2 * LV_MAX + i] = a.z; buf[3 * LV_MAX + i] = a.w;
        buf[4 * LV_MAX + i] = b.x; buf[5 * LV_MAX + i] = b.y; buf[6 * LV_MAX + i] = b.z; buf[7 * LV_MAX + i] = b.w;
    }
    __syncthreads();
    stamp();
    const hashq::Lane lane = hashq::make_lane(tid);
    const hashx::Lane row = hashx::make_lane(tid);
    uint32_t base = 0;
    size_t off = 0;
    for (uint32_t cnt = chunk; cnt > 1; cnt >>= 1) {
        const uint32_t half = cnt >> 1;
        if (half <= 16) {
            const uint32_t node = tid >> 4, w = tid & 15u, j = w >> 2;
            if (node < half) {
                const uint32_t *src = buf + base + 2 * node;
                const uint32_t ml = hashx::message(src[j * LV_MAX], src[(4 + j) * LV_MAX], row);
                const uint32_t mr = hashx::message(src[j * LV_MAX + 1], src[(4 + j) * LV_MAX + 1], row);
                const uint32_t x = hashx::node_hash(ml, mr, row);
                uint8_t *dst = (uint8_t *)(nodes + 2 * (off + node));
                dst[w] = (uint8_t)x; dst[16 + w] = (uint8_t)(x >> 16);
                uint8_t *nb = (uint8_t *)(buf + (base ^ (LV_MAX / 2)) + node);
                nb[4 * (j * LV_MAX) + (w & 3u)] = (uint8_t)x;
                nb[4 * ((4 + j) * LV_MAX) + (w & 3u)] = (uint8_t)(x >> 16);
            }
        } else {
            const uint32_t node = tid >> 2;
            if (node < half) {
                uint32_t l[8], r[8], lo, hi;
                for (int w = 0; w < 8; w++) { l[w] = buf[w * LV_MAX + base + 2 * node]; r[w] = buf[w * LV_MAX + base + 2 * node + 1]; }
                hashq::node_hash(l, r, lane, lo, hi);
                uint32_t *dst = (uint32_t *)(nodes + 2 * (off + node));
                dst[lane.q] = lo; dst[4 + lane.q] = hi;
                buf[lane.q * LV_MAX + (base ^ (LV_MAX / 2)) + node] = lo;
                buf[(4 + lane.q) * LV_MAX + (base ^ (LV_MAX / 2)) + node] = hi;
            }
        }
        off += half;
        base ^= LV_MAX / 2;
        __syncthreads();
        stamp();
    }
    if (tid == 0) {
        uint32_t m[8];
        for (int w = 0; w < 8; w++) m[w] = buf[w * LV_MAX + base];
        hashc::fs_absorb_root(fs_words, m, nullptr, alpha);
    }
    stamp();
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
        if (chain == 1) {
            hex_kernel<<<(16 * n + 255) / 256, 256>>>(din, db, n, 1);
            (void)hipMemcpy(b.data(), db, b.size() * 4, hipMemcpyDeviceToHost);
            int badx = 0;
            for (int i = 0; i < 8 * n; i++) badx += a[i] != b[i];
            printf("row of sixteen: %d of %d digest words differ\n", badx, 8 * n);
            bad += badx;
        }
    }
    // latency: one wave, a chain of dependent node hashes
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int chain = 200;
    for (int which = 0; which < 3; which++) {
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            if (which == 0) single_kernel<<<1, 64>>>(din, da, 64, chain);
            else if (which == 1) quad_kernel<<<1, 64>>>(din, db, 16, chain);
            else hex_kernel<<<1, 64>>>(din, db, 4, chain);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
        }
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f us per node hash in a single wave\n", which == 0 ? "single lane   " : which == 1 ? "quad lanes    " : "row of sixteen", 1e3 * ms / chain);
    }
    uint4 *dn;
    (void)hipMalloc(&dn, (size_t)400 * 64 * 32);
    {
        uint32_t *dfs; uint64_t *dal, *dst;
        (void)hipMalloc(&dfs, 64); (void)hipMalloc(&dal, 8); (void)hipMalloc(&dst, 8 * 32);
        (void)hipMemset(dfs, 1, 64);
        for (uint32_t chunk : {256u, 64u, 512u}) {
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                (void)hipEventRecord(e0);
                final_kernel<<<1, 1024>>>(din, dn, chunk, dfs, dal, dst);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            uint64_t st[32];
            (void)hipMemcpy(st, dst, sizeof st, hipMemcpyDeviceToHost);
            int levels = 0;
            while ((1u << levels) < chunk) levels++;
            printf("last launch of a tree, %u digests -> root + Fiat-Shamir: %.1f us between events; load %.2f us, levels", chunk, 1e3 * ms, (st[1] - st[0]) * 0.01);
            for (int l = 0; l < levels; l++) printf(" %.2f", (st[2 + l] - st[1 + l]) * 0.01);
            printf(", Fiat-Shamir round %.2f us, sum %.2f us\n", (st[2 + levels] - st[1 + levels]) * 0.01, (st[2 + levels] - st[0]) * 0.01);
        }
    }
    for (uint32_t half : {64u, 32u, 16u, 4u, 1u}) {
        float t[2];
        for (int mode = 0; mode < 2; mode++) {
            const int levels = 400;
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(e0);
                if (mode == 0) level_kernel<0><<<1, 1024>>>(din, dn, half, levels);
                else level_kernel<1><<<1, 1024>>>(din, dn, half, levels);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
            }
            (void)hipEventElapsedTime(&t[mode], e0, e1);
            t[mode] = 1e3f * t[mode] / levels;
        }
        printf("level of %2u nodes in a 1024-lane workgroup: quad %.2f us, row of sixteen %.2f us\n", half, t[0], t[1]);
    }
    return bad != 0;
}
