// Microbenchmark (run on the GPU box): what does a random narrow read cost as a function of how much of its 128-B line is
// touched?  N random "cells" of a table far larger than the Infinity Cache; per cell the kernel loads
//   mode 0: 4 B at +0            mode 1: 16 B at +0          mode 2: 16 B at +0 and +64
//   mode 3: 64 B (+0..+63)       mode 4: the whole 128 B     mode 5: 32 B (+0..+31)
// Cells are 128-B aligned (stride 128) or 64-B aligned (stride 64, modes 0,1,3,5).  Reports M cells/s and the FETCH_SIZE-free
// byte rate; rocprofv3 --pmc FETCH_SIZE over this binary calibrates the counter for narrow reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ table, const uint32_t* __restrict__ idx, uint64_t n,
                                                uint32_t stride_words, int mode, uint32_t* __restrict__ out)
{
    const uint64_t i0 = (uint64_t(blockIdx.x) * 256 + threadIdx.x) * 4;
    uint32_t acc = 0;
    uint4 v[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t i = i0 + j < n ? i0 + j : 0;
        const uint32_t* p = table + uint64_t(idx[i]) * stride_words;
        v[j][0] = make_uint4(0, 0, 0, 0); v[j][1] = make_uint4(0, 0, 0, 0);
        if (mode == 0) v[j][0].x = p[0];
        else if (mode == 1) v[j][0] = *reinterpret_cast<const uint4*>(p);
        else if (mode == 2) { v[j][0] = *reinterpret_cast<const uint4*>(p); v[j][1] = *reinterpret_cast<const uint4*>(p + 16); }
        else if (mode == 5) { v[j][0] = *reinterpret_cast<const uint4*>(p); v[j][1] = *reinterpret_cast<const uint4*>(p + 4); }
        else {
            const int nw = mode == 3 ? 4 : 8;
            for (int w = 0; w < nw; ++w) { uint4 t = *reinterpret_cast<const uint4*>(p + 4 * w); v[j][w & 1].x ^= t.x ^ t.y ^ t.z ^ t.w; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc ^= v[j][0].x ^ v[j][0].y ^ v[j][0].z ^ v[j][0].w ^ v[j][1].x ^ v[j][1].y ^ v[j][1].z ^ v[j][1].w;
    if (i0 < n) out[i0 / 4] = acc;
}

// The access pattern of k_fill on a cells layout: G consecutive lanes read the G consecutive words of one cell (one coalesced
// request of 4 G bytes per cell), four cells per group in flight.
__global__ __launch_bounds__(256) void k_gather_coop(const uint32_t* __restrict__ table, const uint32_t* __restrict__ idx, uint64_t n,
                                                     uint32_t stride_words, uint32_t G, uint32_t* __restrict__ out)
{
    const uint64_t t = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    const uint64_t g = t / G, j = t % G;
    uint32_t acc = 0, v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint64_t i = g * 4 + c < n ? g * 4 + c : 0;
        v[c] = table[uint64_t(idx[i]) * stride_words + j];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc ^= v[c];
    if (g * 4 < n) out[t % (n / 4 + 1)] = acc;
}

int main(int argc, char** argv)
{
    const uint64_t table_bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096ull) << 20;   // MiB
    const uint64_t n = argc > 2 ? strtoull(argv[2], nullptr, 10) : 12500000ull;
    uint32_t* d_table; uint32_t* d_idx; uint32_t* d_out;
    hipMalloc(&d_table, table_bytes + 256);
    hipMemset(d_table, 1, table_bytes + 256);
    hipMalloc(&d_idx, n * 4);
    hipMalloc(&d_out, (n / 4 + 1) * 4);
    std::vector<uint32_t> h(n);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (uint32_t stride : {128u, 64u, 32u}) {
        const uint64_t cells = table_bytes / stride;
        uint64_t z = 0x9E3779B97F4A7C15ull;
        for (uint64_t i = 0; i < n; ++i) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; h[i] = uint32_t(z % cells); }
        hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int mode : {0, 1, 5, 2, 3, 4}) {
            if (stride == 64 && (mode == 2 || mode == 4)) continue;
            if (stride == 32 && (mode == 2 || mode == 3 || mode == 4)) continue;
            const unsigned blocks = unsigned((n / 4 + 255) / 256);
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, d_table, d_idx, n, stride / 4, mode, d_out);
            hipEventRecord(a);
            const int reps = 5;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, d_table, d_idx, n, stride / 4, mode, d_out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
            const int bytes = mode == 0 ? 4 : mode == 1 ? 16 : mode == 2 ? 32 : mode == 3 ? 64 : mode == 5 ? 32 : 128;
            printf("table %llu MiB stride %3u mode %d (%3d B/cell): %.4f ms  %.0f M cells/s  useful %.2f TB/s  if-128B-lines %.2f TB/s  if-64B %.2f TB/s\n",
                   (unsigned long long)(table_bytes >> 20), stride, mode, bytes, ms, n / ms / 1e3, n * double(bytes) / ms / 1e9,
                   n * 128.0 / ms / 1e9, n * 64.0 / ms / 1e9);
        }
        // cooperative reads: G lanes per cell (modes 10 + log2 G)
        for (uint32_t G : {8u, 16u, 32u}) {
            if (G * 4 > stride) continue;
            const unsigned blocks = unsigned((n / 4 * G + 255) / 256);
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_gather_coop, dim3(blocks), dim3(256), 0, 0, d_table, d_idx, n, stride / 4, G, d_out);
            hipEventRecord(a);
            const int reps = 5;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_gather_coop, dim3(blocks), dim3(256), 0, 0, d_table, d_idx, n, stride / 4, G, d_out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
            const int bytes = int(G * 4);
            printf("table %llu MiB stride %3u mode %d (%3d B/cell): %.4f ms  %.0f M cells/s  useful %.2f TB/s  if-128B-lines %.2f TB/s  if-64B %.2f TB/s  [%u lanes x 4 B]\n",
                   (unsigned long long)(table_bytes >> 20), stride, G == 8 ? 13 : G == 16 ? 14 : 15, bytes, ms, n / ms / 1e3, n * double(bytes) / ms / 1e9,
                   n * 128.0 / ms / 1e9, n * 64.0 / ms / 1e9, G);
        }
    }
    return 0;
}
