// Device construction of one element whose key space is too large for a histogram (sigma^k > 2^26): the work of
// kmer_index_element::create (kmer_index.hpp:154-179) as
//   hash every k-mer -> (hash, position) pairs -> stable radix sort by hash (positions stay ascending inside a
//   key, the order push_back yields at :160-167) -> heads of the runs -> distinct keys + offsets -> open-addressing
//   slots claimed with one 64-bit compare-and-swap each.
// The sort is a least-significant-digit radix sort written here (8-bit digits, per pass: per-tile digit histograms -> one
// exclusive scan over the digit-major (digit, tile) counts -> stable scatter: round by round inside a tile, wave by wave
// inside a round, lane order inside a wave through ballot matching), over exactly the bits sigma^k needs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>

#include "kmx_kernels.h"

namespace kmx {

namespace {

constexpr unsigned int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_sparse_pairs(const uint8_t* __restrict__ text, uint64_t npos, uint32_t k, uint32_t sigma,
                                                         uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint64_t i = uint64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= npos) return;
    uint64_t h = 0;
    for (uint32_t j = 0; j < k; ++j) h = h * sigma + text[i + j];      // kmer_index.hpp:56-73
    keys[i] = h;
    vals[i] = uint32_t(i);
}

__global__ __launch_bounds__(kBlock) void k_sparse_heads(const uint64_t* __restrict__ keys, uint64_t npos, uint32_t* __restrict__ head)
{
    const uint64_t i = uint64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= npos) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(kBlock) void k_sparse_compact(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ head,
                                                           const uint64_t* __restrict__ rank, uint64_t npos, uint64_t n_ukeys,
                                                           uint64_t* __restrict__ ukeys, uint32_t* __restrict__ offs)
{
    const uint64_t i = uint64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= npos) return;
    if (head[i]) {
        const uint64_t j = rank[i];
        ukeys[j] = keys[i];
        offs[j] = uint32_t(i);
    }
    if (i == 0) offs[n_ukeys] = uint32_t(npos);
}

__device__ __forceinline__ uint64_t slot_hash_build(uint64_t key, uint32_t log2cap)
{
    return log2cap ? (key * 0x9E3779B97F4A7C15ull) >> (64 - log2cap) : 0;   // == kmx::slot_hash (kmx_host.h)
}

// Linear probing; a slot is claimed by swapping its {off, cnt} word from 0 (cnt == 0 marks an empty slot, every
// key owns >= 1 position).  Keys are distinct, so an inserter never has to read another slot's key.
__global__ __launch_bounds__(kBlock) void k_sparse_slots(const uint64_t* __restrict__ ukeys, const uint32_t* __restrict__ offs,
                                                         uint64_t n_ukeys, uint32_t log2cap, KmxSlot* __restrict__ slots)
{
    const uint64_t j = uint64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (j >= n_ukeys) return;
    const uint64_t key = ukeys[j];
    const uint32_t off = offs[j], cnt = offs[j + 1] - off;
    const unsigned long long mine = (unsigned long long)off | ((unsigned long long)cnt << 32);
    const uint64_t mask = (uint64_t(1) << log2cap) - 1;
    uint64_t s = slot_hash_build(key, log2cap);
    for (uint64_t probes = 0; probes <= mask; ++probes) {
        unsigned long long* word = reinterpret_cast<unsigned long long*>(&slots[s].off);
        if (atomicCAS(word, 0ull, mine) == 0ull) {
            slots[s].key = key;
            return;
        }
        s = (s + 1) & mask;
    }
}

// ---- LSD radix sort of (u64 key, u32 value) pairs, stable ----
constexpr unsigned int kSortItems = 16;                       // elements per thread and tile
constexpr unsigned int kSortTile = kBlock * kSortItems;       // 4096 per workgroup

__global__ __launch_bounds__(kBlock) void k_rs_hist(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift,
                                                    uint32_t* __restrict__ tile_hist, uint32_t n_tiles)
{
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = uint64_t(blockIdx.x) * kSortTile;
#pragma unroll 4
    for (unsigned int r = 0; r < kSortItems; ++r) {
        const uint64_t i = base + uint64_t(r) * kBlock + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    tile_hist[uint64_t(threadIdx.x) * n_tiles + blockIdx.x] = h[threadIdx.x];     // digit-major: one scan orders (digit, tile)
}

__global__ __launch_bounds__(kBlock) void k_rs_scatter(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t n,
                                                       uint32_t shift, const uint64_t* __restrict__ tile_off, uint32_t n_tiles,
                                                       uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
    constexpr unsigned int NW = kBlock / 64;
    __shared__ unsigned long long goff[256];                   // where this tile's elements of a digit go
    __shared__ unsigned int run[256];                           // ... of which the earlier rounds have placed this many
    __shared__ unsigned int wcnt[NW][256];                      // elements of a digit per wave in the current round
    const unsigned int tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    goff[tid] = tile_off[uint64_t(tid) * n_tiles + blockIdx.x];
    run[tid] = 0;
    const uint64_t base = uint64_t(blockIdx.x) * kSortTile;
    const uint64_t below = (uint64_t(1) << lane) - 1;
    for (unsigned int r = 0; r < kSortItems; ++r) {
#pragma unroll
        for (unsigned int w = 0; w < NW; ++w) wcnt[w][tid] = 0;
        __syncthreads();
        const uint64_t i = base + uint64_t(r) * kBlock + tid;
        const bool valid = i < n;
        const uint64_t key = valid ? keys_in[i] : 0;
        const uint32_t val = valid ? vals_in[i] : 0u;
        const uint32_t dgt = uint32_t(key >> shift) & 255u;
        // the lanes of this wave that hold the same digit (ballot matching over the 8 digit bits)
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dgt >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const uint32_t rank = uint32_t(__popcll(same & below));
        if (valid && rank == 0) wcnt[wv][dgt] = uint32_t(__popcll(same));
        __syncthreads();
        uint32_t pre = 0, tot = 0;
#pragma unroll
        for (unsigned int w = 0; w < NW; ++w) {
            const uint32_t c = wcnt[w][dgt];
            pre += w < wv ? c : 0u;
            tot += c;
        }
        if (valid) {
            const unsigned long long dst = goff[dgt] + run[dgt] + pre + rank;
            keys_out[dst] = key;
            vals_out[dst] = val;
        }
        __syncthreads();                                        // every thread has read run[] and wcnt[]
        if (valid && rank == 0 && pre == 0) run[dgt] += tot;    // (the first wave that holds the digit)
    }
}

struct Temp {
    void* p = nullptr;
    ~Temp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes + 64); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

inline unsigned int blocks(uint64_t n) { return (unsigned int)((n + kBlock - 1) / kBlock); }

} // namespace

// (hash, position) pairs of every k-mer, stably sorted by hash: positions -> d_positions, sorted hashes -> keys_b
static hipError_t sort_pairs(hipStream_t s, const uint8_t* d_text, uint64_t npos, uint32_t k, uint32_t sigma, uint32_t key_bits,
                             uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals, uint32_t* d_positions)
{
    const uint32_t passes = std::max<uint32_t>(1, (key_bits + 7) / 8);
    const uint32_t n_tiles = uint32_t((npos + kSortTile - 1) / kSortTile);
    Temp hist, offs, bsum;
    hipError_t e = hist.alloc(size_t(256) * n_tiles * 4);
    if (e == hipSuccess) e = offs.alloc((size_t(256) * n_tiles + 1) * 8);
    if (e == hipSuccess) e = bsum.alloc((scan_blocks(uint64_t(256) * n_tiles) + 2) * 8);
    if (e != hipSuccess) return e;
    unsigned long long* d_total = reinterpret_cast<unsigned long long*>(bsum.as<uint64_t>() + scan_blocks(uint64_t(256) * n_tiles));
    // the passes ping-pong between (keys_a, vals) and (keys_b, d_positions); the pairs start where an odd number of hops ends in b
    uint64_t* kin = (passes & 1u) ? keys_a : keys_b;
    uint32_t* vin = (passes & 1u) ? vals : d_positions;
    uint64_t* kout = (passes & 1u) ? keys_b : keys_a;
    uint32_t* vout = (passes & 1u) ? d_positions : vals;
    hipLaunchKernelGGL(k_sparse_pairs, dim3(blocks(npos)), dim3(kBlock), 0, s, d_text, npos, k, sigma, kin, vin);
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t shift = 8 * p;
        hipLaunchKernelGGL(k_rs_hist, dim3(n_tiles), dim3(kBlock), 0, s, kin, npos, shift, hist.as<uint32_t>(), n_tiles);
        launch_scan(s, hist.as<uint32_t>(), uint64_t(256) * n_tiles, bsum.as<uint64_t>(), offs.as<uint64_t>(), d_total);
        hipLaunchKernelGGL(k_rs_scatter, dim3(n_tiles), dim3(kBlock), 0, s, kin, vin, npos, shift, offs.as<uint64_t>(), n_tiles, kout, vout);
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);                      // the temporaries are released on return
    return e;
}

hipError_t sort_kmer_positions(hipStream_t s, const uint8_t* d_text, uint64_t n, uint32_t k, uint32_t sigma, uint32_t key_bits,
                               uint32_t* d_positions)
{
    const uint64_t npos = n - k + 1;
    Temp keys_a, keys_b, vals;
    hipError_t e = keys_a.alloc(npos * 8);
    if (e == hipSuccess) e = keys_b.alloc(npos * 8);
    if (e == hipSuccess) e = vals.alloc(npos * 4);
    if (e != hipSuccess) return e;
    return sort_pairs(s, d_text, npos, k, sigma, key_bits, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals.as<uint32_t>(), d_positions);
}

hipError_t build_sparse_element(hipStream_t s, const uint8_t* d_text, uint64_t n, uint32_t k, uint32_t sigma, uint32_t key_bits,
                                uint32_t* d_positions, SparseTables* out)
{
    static_assert(sizeof(KmxSlot) == 16, "slot layout: {u64 key, u32 off, u32 cnt}");
    *out = SparseTables{};
    const uint64_t npos = n - k + 1;
    Temp keys_a, keys_b, vals, bsum;
    hipError_t e = keys_a.alloc((npos + 1) * 8);
    if (e == hipSuccess) e = keys_b.alloc((npos + 1) * 8);
    if (e == hipSuccess) e = vals.alloc(npos * 4);
    if (e == hipSuccess) e = bsum.alloc((scan_blocks(npos) + 1) * 8);                // block sums + the scan's total
    if (e != hipSuccess) return e;
    e = sort_pairs(s, d_text, npos, k, sigma, key_bits, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals.as<uint32_t>(), d_positions);
    if (e != hipSuccess) return e;

    // heads of the runs -> index of every distinct key (vals and keys_a are free again)
    uint32_t* head = vals.as<uint32_t>();
    uint64_t* rank = keys_a.as<uint64_t>();
    unsigned long long* d_total = reinterpret_cast<unsigned long long*>(bsum.as<uint64_t>() + scan_blocks(npos));
    hipLaunchKernelGGL(k_sparse_heads, dim3(blocks(npos)), dim3(kBlock), 0, s, keys_b.as<uint64_t>(), npos, head);
    launch_scan(s, head, npos, bsum.as<uint64_t>(), rank, d_total);
    unsigned long long n_ukeys = 0;
    e = hipMemcpyAsync(&n_ukeys, d_total, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;

    uint32_t log2cap = 4;
    while ((uint64_t(1) << log2cap) < 2 * n_ukeys) ++log2cap;               // load <= 0.5, as build_slots (kmx_host.cpp)
    const uint64_t cap = uint64_t(1) << log2cap;
    void *p_ukeys = nullptr, *p_offs = nullptr, *p_slots = nullptr;
    e = hipMalloc(&p_ukeys, n_ukeys * 8 + 64);
    if (e == hipSuccess) e = hipMalloc(&p_offs, (n_ukeys + 1) * 4 + 64);
    if (e == hipSuccess) e = hipMalloc(&p_slots, cap * sizeof(KmxSlot) + 64);
    if (e == hipSuccess) e = hipMemsetAsync(p_slots, 0, cap * sizeof(KmxSlot), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sparse_compact, dim3(blocks(npos)), dim3(kBlock), 0, s, keys_b.as<uint64_t>(), head, rank, npos, uint64_t(n_ukeys),
                           static_cast<uint64_t*>(p_ukeys), static_cast<uint32_t*>(p_offs));
        hipLaunchKernelGGL(k_sparse_slots, dim3(blocks(n_ukeys)), dim3(kBlock), 0, s, static_cast<const uint64_t*>(p_ukeys),
                           static_cast<const uint32_t*>(p_offs), uint64_t(n_ukeys), log2cap, static_cast<KmxSlot*>(p_slots));
        e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e != hipSuccess) {
        if (p_ukeys) (void)hipFree(p_ukeys);
        if (p_offs) (void)hipFree(p_offs);
        if (p_slots) (void)hipFree(p_slots);
        return e;
    }
    out->d_ukeys = static_cast<uint64_t*>(p_ukeys);
    out->d_offs = static_cast<uint32_t*>(p_offs);
    out->d_slots = static_cast<KmxSlot*>(p_slots);
    out->n_ukeys = n_ukeys;
    out->log2cap = log2cap;
    return hipSuccess;
}

} // namespace kmx
