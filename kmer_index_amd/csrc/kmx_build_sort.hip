// Device construction of one element whose key space is too large for a histogram (sigma^k > 2^26): the work of
// kmer_index_element::create (kmer_index.hpp:154-179) as
//   hash every k-mer -> (hash, position) pairs -> stable radix sort by hash (positions stay ascending inside a
//   key, the order push_back yields at :160-167) -> heads of the runs -> distinct keys + offsets -> open-addressing
//   slots claimed with one 64-bit compare-and-swap each.
// The sort itself is rocPRIM's device radix sort (vendor library for a plain library step, not on the search
// path); everything around it is written here.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

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
    hipLaunchKernelGGL(k_sparse_pairs, dim3(blocks(npos)), dim3(kBlock), 0, s, d_text, npos, k, sigma, keys_a, vals);
    Temp sort_tmp;
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a, keys_b, vals, d_positions, size_t(npos), 0u, key_bits, s);
    if (e == hipSuccess) e = sort_tmp.alloc(tmp_bytes);
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(sort_tmp.p, tmp_bytes, keys_a, keys_b, vals, d_positions, size_t(npos), 0u, key_bits, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);                      // sort_tmp is released on return
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
