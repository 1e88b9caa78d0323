// Launch interface of the gfx950 kernels (kmx_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kmx_types.h"


namespace kmx {

// Per-query descriptor arrays (structure of arrays, all device pointers).
struct QueryDesc {
    uint64_t* src;          // arena index of the run that feeds the query (bucket / prefix slice / candidates)
    uint32_t* cnt;          // number of hits (STITCH: written by k_validate)
    uint32_t* c0;           // STITCH: number of candidates; PREFIX: number of runs in the slice
    uint64_t* aux;          // STITCH: index of the first mask word; PREFIX: last-kmer bits (bit j <-> position n-j)
    uint64_t* key;          // PREFIX: key index of the first run (offs + key bounds the runs); STITCH: arena index of that further part's bucket
    uint64_t* p1;           // STITCH: one further part, (its offset in the query << 32) | its bucket size; bit 63: more parts follow
    uint8_t* kind;          // kmx_query_kind
    uint8_t* status;        // kmx_query_status
    uint32_t* stitch_list;  // indices of the STITCH queries (arbitrary order)
    uint32_t* prefix_list;  // indices of the PREFIX queries (arbitrary order)
    uint32_t* short_list;   // indices of the STITCH queries of the short class (KMX_VSHORT; k_validate_short), from its front
    uint32_t* stitch_hits;  // STITCH: surviving candidates of query q at [64 * aux[q], 64 * aux[q] + cnt[q]), written by
                            // k_validate and copied out by k_fill; nullptr -> k_compact decodes the masks instead
};

// k_lookup variants: items = queries per thread (4 or 8), pairs = the interleaved pass takes two-part cross-referenced queries
// (pairs implies items == 4).  block_hits receives lookup_blocks(nq, items) sums: the first level of the scan behind it.
uint64_t lookup_blocks(uint64_t nq, int items);
void launch_lookup(hipStream_t s, int items, bool pairs, const KmxIndexDev* ix, const uint8_t* qranks, const uint64_t* qoff,
                   uint64_t nq, const QueryDesc& d, unsigned long long* ctr, uint64_t* block_hits, uint32_t flags);
// behind launch_lookup with KMX_SEARCH_INTERNAL_DEFER_LONG in its flags: the queries of very many parts it listed, a wave each
void launch_lookup_long(hipStream_t s, const KmxIndexDev* ix, const uint8_t* qranks, const uint64_t* qoff, uint64_t nq, const QueryDesc& d,
                        unsigned long long* ctr, uint32_t flags);
void launch_validate(hipStream_t s, const KmxIndexDev* ix, const uint32_t* arena, const uint8_t* qranks, const uint64_t* qoff,
                     const QueryDesc& d, uint64_t n_stitch, uint64_t n_more, uint64_t n_tiny, const uint32_t* tiny_list, uint64_t n_short,
                     uint64_t* mask_words, bool direct);   // direct: tiny queries whose survivors are one run of the first bucket leave as plain copies
// the whole search of a small batch in one launch; `mailbox` is page-locked host memory (layout: kmx_types.h)
// n_blocks workgroups; xchg: n_blocks zeroed u64 words of device memory (the workgroups' totals), may be NULL for one workgroup
void launch_small(hipStream_t s, const KmxIndexDev* ix, const uint32_t* arena, unsigned char* mailbox, const KmxSmallLayout& layout, uint32_t n_blocks,
                  const KmxSmallArgs& args, uint32_t nq_total, unsigned long long* xchg, uint32_t flags);
uint64_t scan_blocks(uint64_t n);
void launch_build_dir(hipStream_t s, const uint64_t* d_ukeys, uint64_t n_ukeys, uint32_t shift, uint32_t n_dir, uint32_t* d_dir);
// cells of a dense element (KmxElemDev::cnt8): d_region = the element's contiguous copy, d_cells = cell 0
void launch_build_cells(hipStream_t s, const uint32_t* d_offs, const uint32_t* d_region, uint64_t n_keys, uint32_t cell_shift,
                        uint32_t* d_cells, uint8_t* d_cnt8);

// kmx_build_sort.hip — device construction of an element with a key space beyond the histogram path:
// positions (grouped by hash, ascending inside a group) into d_positions[n - k + 1], tables into new allocations
// the caller owns.  key_bits = bits needed for sigma^k - 1 (the radix sort skips the rest).
struct SparseTables {
    uint64_t* d_ukeys = nullptr;   // n_ukeys distinct hashes, ascending
    uint32_t* d_offs = nullptr;    // n_ukeys + 1 group boundaries
    KmxSlot* d_slots = nullptr;    // 1 << log2cap open-addressing slots
    uint64_t n_ukeys = 0;
    uint32_t log2cap = 0;
};
// only the positions, grouped by hash and ascending inside a group (for an element whose tables come from the
// histogram but whose buckets are too long for the LDS sorts)
hipError_t sort_kmer_positions(hipStream_t s, const uint8_t* d_text, uint64_t n, uint32_t k, uint32_t sigma, uint32_t key_bits,
                               uint32_t* d_positions);
hipError_t build_sparse_element(hipStream_t s, const uint8_t* d_text, uint64_t n, uint32_t k, uint32_t sigma, uint32_t key_bits,
                                uint32_t* d_positions, SparseTables* out);
void launch_scan(hipStream_t s, const uint32_t* in, uint64_t n, uint64_t* bsum, uint64_t* out,
                 unsigned long long* total_out);
// k_fill build variants: e = output slots per thread (tile = 256 * e), nt = non-temporal stores
// of the hit lists.
struct FillVariant {
    int e;
    bool nt;
};
// Counter publication folded into the scan (steady state: no memset and no copy operation on the stream): the block that
// writes the grand total also copies the batch's counter block `cur` to page-locked host memory `host` and zeroes `next`,
// the block the next batch on this result handle counts into.  All NULL: nothing of the kind.
struct CounterPub {
    const unsigned long long* cur = nullptr;
    unsigned long long* next = nullptr;
    unsigned long long* host = nullptr;
};
// returns true when the publication was done by the scan (the fused-spine path), false when the caller has to copy / reset
bool launch_scan_tiles(hipStream_t s, const uint32_t* in, uint64_t n, uint64_t* bsum, uint64_t* out,
                       unsigned long long* total_out, uint64_t tile, uint64_t n_tiles_cap, uint32_t* tile_q, int lookup_items,
                       const CounterPub& pub = CounterPub());   // lookup_items: bsum holds the sums of the k_lookup variant with that many queries per thread (0: none)
FillVariant effective_fill_variant(const FillVariant& v, bool rec32);
uint64_t fill_tile(const FillVariant& v);
void launch_partition(hipStream_t s, const uint64_t* off, uint64_t nq, uint64_t tile, uint64_t n_tiles, uint32_t* tile_q);
void launch_fill(hipStream_t s, const FillVariant& v, bool rec32, const KmxIndexDev* ix, const uint32_t* arena, const uint64_t* hit_off,
                 const uint32_t* tile_q, const unsigned long long* total_dev, uint64_t n_tiles, const QueryDesc& d, uint32_t* out);
void launch_compact(hipStream_t s, const uint32_t* arena, const QueryDesc& d, uint64_t n_stitch,
                    const uint64_t* mask_words, const uint64_t* hit_off, uint32_t* out);
void launch_prefix_len(hipStream_t s, const QueryDesc& d, uint64_t n_prefix, const uint32_t* banded, uint32_t* plen);
// index construction on the device (see kmx_kernels.hip)
void launch_build_phase1(hipStream_t s, const uint8_t* d_text, uint64_t n, uint32_t k, uint32_t sigma, uint64_t n_keys,
                         uint32_t* d_hist, uint64_t* d_scratch_u64, uint64_t* d_bsum, uint32_t* d_offs, uint32_t* d_cursor,
                         unsigned int* d_info, unsigned long long* d_total);
void launch_build_phase2(hipStream_t s, const uint8_t* d_text, uint64_t n, uint32_t k, uint32_t sigma, uint64_t n_keys,
                         const uint32_t* d_offs, uint32_t* d_hist, uint64_t* d_scratch_u64, uint64_t* d_bsum, uint32_t* d_cursor,
                         unsigned int* d_info, unsigned long long* d_total, uint32_t* d_region, uint32_t* d_aoffs, uint32_t a0,
                         int sort_mode, uint32_t* d_atab, uint32_t region_end);
void launch_bucket_sort_block(hipStream_t s, const uint32_t* d_offs, uint64_t n_keys, uint32_t* d_positions);
// the PREFIX queries with a short slice (front of prefix_list): two kernels, each takes its class and skips the other's
void launch_prefix_sort_small(hipStream_t s, const KmxIndexDev* ix, const uint64_t* qoff, const QueryDesc& d, uint64_t n_prefix,
                              const uint64_t* hit_off, const uint32_t* arena, uint32_t* out);
void launch_prefix_merge_small(hipStream_t s, const KmxIndexDev* ix, const uint64_t* qoff, const QueryDesc& d, uint64_t n_prefix,
                               const uint64_t* hit_off, const uint32_t* arena, uint32_t* out);
// tile_off / tmp: only when the batch has slices beyond KMX_PSORT_BLOCK_CAP (launch_prefix_len + a scan give tile_off)
// n_mid: how many of the listed slices have at most KMX_PSORT_MID_CAP positions (they take the 256-thread shape of the kernel)
// The slices beyond the 256-thread shape that can be cut into BANDS (value ranges of the text: kmx_kernels.hip): banded[i] = 1 for
// them, 0 for every other listed slice.  bands: cap_bands records of prefix_item_bytes() each, cap_bands >= 6 per listed slice beyond
// KMX_PSORT_MID_CAP + (positions of the slices beyond KMX_PSORT_BLOCK_CAP) / prefix_band_target(); cuts: cap_cuts words (a slice that
// finds no room stays with the chunks); used: two zeroed device counters (KMX_CTR_PSB_BANDS, KMX_CTR_PSB_CUTS)
// ... and the slices beyond one chunk with too many runs for bands are SPREAD BY VALUE (banded[i] = 2): room for their records
// (prefix_split_bytes(0) each, cleared), their tiles (prefix_split_bytes(1) each, filled with 0xFF), their counters (words, cleared) and
// their positions (words); splits == nullptr: none.  used: SIX zeroed device counters (KMX_CTR_PSB_BANDS ...)
struct PrefixSplitRoom {
    void* splits; uint64_t cap_splits;       // >= slices beyond KMX_PSORT_BLOCK_CAP
    void* tiles; uint64_t cap_tiles;         // >= their positions / prefix_split_tile() + their number
    uint32_t* counters; uint64_t cap_counters;   // >= 3 * (their positions / prefix_split_target() + their number)
    uint32_t* scratch; uint64_t cap_scratch;     // >= their positions + 4 * their number
};
void launch_prefix_bands(hipStream_t s, const KmxIndexDev* ix, const uint64_t* qoff, const QueryDesc& d, uint64_t n_prefix, const uint64_t* hit_off,
                         const uint32_t* arena, uint32_t* banded, void* bands, uint64_t cap_bands, uint32_t* cuts, uint64_t cap_cuts,
                         unsigned long long* used, const PrefixSplitRoom& sr);
void launch_prefix_split(hipStream_t s, const PrefixSplitRoom& sr, const unsigned long long* used, const uint32_t* arena, uint32_t* banded, uint64_t n_text,
                         void* items, uint64_t cap_items, unsigned long long* n_other);
uint64_t prefix_split_bytes(int what);
uint64_t prefix_split_target();
uint64_t prefix_split_tile();
// items: room for cap_items records of prefix_item_bytes() each, cap_items >= (listed slices beyond KMX_PSORT_MID_CAP) + (positions of the
// slices beyond KMX_PSORT_BLOCK_CAP) / KMX_PSORT_BLOCK_CAP; n_items: two zeroed device counters (KMX_CTR_PSB_MERGE, KMX_CTR_PSB_OTHER);
// banded / bands / cuts / n_bands: what launch_prefix_bands left; split: the scratch buffer of launch_prefix_split; mid_items: n_prefix records of
// prefix_item_bytes() (only when n_mid > 0); dbg: the index's debug words
void launch_prefix_sort_block(hipStream_t s, const KmxIndexDev* ix, const uint64_t* qoff, const QueryDesc& d, uint64_t n_prefix, uint64_t n_mid,
                              const uint64_t* hit_off, const uint32_t* arena, uint32_t* out, const uint64_t* tile_off, uint32_t* tmp,
                              void* items, uint64_t cap_items, unsigned long long* n_items, const uint32_t* banded, const void* bands, uint64_t cap_bands,
                              const uint32_t* cuts, const unsigned long long* n_bands, const uint32_t* split, void* mid_items, uint64_t n_text,
                              unsigned long long* dbg);
uint64_t prefix_item_bytes();
uint64_t prefix_band_target();
uint64_t prefix_band_min();
uint64_t prefix_band_runs();
// one pairwise merge pass over the sorted chunks of the large slices; max_tiles >= tile_off[n_prefix]
void launch_prefix_merge_pass(hipStream_t s, const QueryDesc& d, uint64_t n_prefix, const uint64_t* tile_off, uint64_t max_tiles,
                              const uint64_t* hit_off, uint32_t* out, uint32_t* tmp, uint32_t pass);
uint64_t prefix_merge_tile();
// prefix levels: the batch of all m-mers in rank-hash order (m * nq letters, nq + 1 offsets); 64-bit offsets as a 32-bit table
void launch_all_kmers(hipStream_t s, uint32_t m, uint32_t sigma, uint64_t nq, uint8_t* d_qranks, uint64_t* d_qoff);
void launch_narrow_offsets(hipStream_t s, const uint64_t* d_in, uint64_t n, uint32_t* d_out);
// off[i] += add for i < n (kmx_result_gather_device: a part's hit_off entries behind the hits of the parts in front of it)
void launch_rebase_offsets(hipStream_t s, uint64_t* d_off, uint64_t n, uint64_t add);

} // namespace kmx
