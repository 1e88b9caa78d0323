// Host-side pieces of the engine: fast_pow, the planner (choose_search_scheme)
// and the flatten of one text into per-k position arrays + tables.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "kmx_types.h"

namespace kmx {

uint64_t fast_pow(uint64_t base, uint8_t exp);
// sigma^k exactly, for a valid k (k_is_valid: the product fits 64 bits).  NOT fast_pow: the reference's fast_pow
// returns 0 from exp == 63 on (fast_pow.hpp:19), which is what (sigma, k) = (2, 63) would ask it for.
uint64_t key_space(uint32_t sigma, uint32_t k);

// kmer_index::choose_search_scheme (kmer_index.hpp:407-476).
struct Plan {
    std::vector<uint8_t> use_multi;               // _use_multi_search_scheme
    std::vector<std::vector<uint32_t>> nk_sum;    // _optimal_nk_sum
};
Plan make_plan(const std::vector<uint32_t>& ks, uint32_t range);
// Device form of the plan for an index holding `ks` (template order).
std::vector<KmxPlanEntry> make_plan_entries(const std::vector<uint32_t>& ks, uint32_t range);
// ... and the engine's variant of it for searches that need not reproduce the reference's result object: long single-k
// queries on the largest k that fits them (kmx_host.cpp)
std::vector<KmxPlanEntry> make_fast_plan_entries(const std::vector<uint32_t>& ks, uint32_t range, uint32_t sigma);

// choose_best_k (choose_best_k.hpp:12-60): which n_k values of k to instantiate for a set of query lengths.
std::vector<uint32_t> choose_best_k(const uint64_t* lengths, uint64_t n_lengths, uint32_t n_k);

// Host image of one flattened kmer_index_element.
struct ElemImage {
    uint32_t k = 0;
    uint32_t table_kind = 0;
    uint32_t log2cap = 0;
    uint64_t n_keys = 0;
    uint64_t npos = 0;
    std::vector<uint32_t> positions;   // region entries: npos grouped by hash [+ the line-aligned copy of the groups]
    std::vector<uint32_t> aoffs;       // aligned copy: start of every group (transient: feeds the open slots / atab)
    std::vector<uint32_t> atab;        // dense: packed (start/32) << 5 | (count & 31), n_keys + 1 entries
    uint64_t region = 0;               // npos, or the end of the aligned copy
    std::vector<uint32_t> offs;        // dense: n_keys + 1; open: ukeys.size() + 1
    std::vector<uint64_t> ukeys;       // open only
    std::vector<KmxSlot> slots;        // open only
    // built on the device (kmx_capi.hip): positions already sit in the arena, dense offs too
    bool positions_on_device = false;
    const uint32_t* d_offs_prebuilt = nullptr;
    const uint32_t* d_atab_prebuilt = nullptr;
    // open table built on the device (kmx_build_sort.hip): d_offs_prebuilt holds n_ukeys_prebuilt + 1 boundaries
    const uint64_t* d_ukeys_prebuilt = nullptr;
    const KmxSlot* d_slots_prebuilt = nullptr;
    uint64_t n_ukeys_prebuilt = 0;
};

// Builds the image of one element — the work of kmer_index_element::create
// (kmer_index.hpp:154-179).  Returns false and sets err on invalid parameters.
bool flatten_element(const uint8_t* ranks, uint64_t n, uint32_t sigma, uint32_t k, uint32_t table_kind,
                     ElemImage& out, std::string& err, bool aligned_copy = true);
// appends the 128-byte-aligned copy of the groups to im.positions (when the buckets are long enough)
void add_aligned_copy(ElemImage& im);

// AUTO -> DENSE when sigma^k <= 4 (n-k+1) (and the key space fits a histogram), else OPEN.
uint32_t resolve_table_kind(uint32_t sigma, uint32_t k, uint64_t n, uint32_t requested);
// open-addressing slots from sorted distinct keys + offsets (im.ukeys, im.offs)
void build_slots(ElemImage& im);

// static_assert(k > 0 and k < 64 / log2(sigma)) of kmer_index.hpp:42-43.
bool k_is_valid(uint32_t sigma, uint32_t k);

inline uint64_t slot_hash(uint64_t key, uint32_t log2cap)
{
    return log2cap ? (key * 0x9E3779B97F4A7C15ull) >> (64 - log2cap) : 0;
}

} // namespace kmx
