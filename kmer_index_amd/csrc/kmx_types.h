// Shared host/device layout of the flattened index image and of the per-batch
// query descriptors.  Everything on the search path is unsigned integer work:
// u64 rank-hashes, u32 text positions, u64 mask words.
#pragma once
#include <stdint.h>
#include "../../include/kmx.h"

#define KMX_MAX_LEVELS 3            // prefix levels per element (KmxElemDev::n_levels)
#define KMX_DEFAULT_PREFIX_LEVELS 2  // kmx_options::prefix_levels == 0 and no KMX_PREFIX_LEVELS in the environment

// One slot of the open-addressing table that replaces
// robin_hood::unordered_map<size_t, std::vector<position_t>> (kmer_index.hpp:52).
// cnt == 0 marks an empty slot (every stored key owns >= 1 position).
struct KmxSlot {
    uint64_t key;
    uint32_t off;   // first position of the key's run, relative to the element's arena_base
    uint32_t cnt;
};

// One kmer_index_element (kmer_index.hpp:39-347) flattened:
//   positions of all k-mers, grouped by rank-hash in ascending hash order and
//   ascending inside a group (the order push_back produces at :160-167), stored
//   in the shared arena at [arena_base, arena_base + npos);
//   for elements with long buckets (>= 32 positions on average) a SECOND copy of the
//   groups follows in which every group starts on a 128-byte line: exact lookups and
//   stitch candidates read that copy (a 381-byte run then touches 3 lines, not 4 on
//   average), prefix ranges keep using the contiguous copy;
//   dense table : offs[h] .. offs[h+1] bound the group of hash h (n_offs = sigma^k + 1);
//   open table  : ukeys[i] (ascending distinct hashes) with offs[i] .. offs[i+1],
//                 plus KmxSlot slots[1 << log2cap] for O(1) exact probes.
struct KmxElemDev {
    uint32_t k;
    uint32_t table_kind;   // KMX_TABLE_OPEN | KMX_TABLE_DENSE
    uint32_t log2cap;      // open: log2 of the slot count
    uint32_t n_ukeys;      // open: number of distinct keys
    uint64_t n_keys;       // sigma^k
    uint64_t arena_base;   // element index into the arena
    uint64_t npos;         // n - k + 1
    const uint32_t* offs;  // group boundaries of the contiguous copy, relative to arena_base
    const KmxSlot* slots;  // open only
    const uint64_t* ukeys; // open only
    const uint32_t* atab;  // dense + line-aligned copy: n_keys + 1 packed entries (start/32) << 5 | (count & 31) —
                           // start of the group in that copy (relative to arena_base, a multiple of 32) and the low
                           // bits of its size; the size itself is the padded difference to the next entry adjusted by
                           // those bits, so an exact lookup reads 8 bytes of ONE 4-byte-per-key table.  Open tables
                           // carry the aligned start in KmxSlot::off instead.  NULL: no aligned copy.
    uint64_t region;       // elements this element occupies in the arena (contiguous copy [+ aligned copy])
    const uint32_t* dir;   // open only (NULL: none): dir[j] = index of the first distinct key >= j << dir_shift, n_dir + 1
                           // entries — narrows the two binary searches of a prefix range from the whole key array to one
                           // directory cell.  Derived from ukeys when the index is installed; not part of the image.
    uint32_t dir_shift;
    uint32_t n_dir;
    // dense tables with SHORT buckets (a handful of positions per key: DNA5 k = 10 on 1e8 bp, protein k = 5): a third
    // layout of the groups, "cells" — key h owns the fixed-size, line-aligned cell [cell_base + (h << cell_shift), ... + 2^cell_shift)
    // of the arena holding its positions (ascending, zero padded), and cnt8[h] its size, 255 for a key whose group does not
    // fit a cell (those are served from offs / the contiguous copy as before).  An exact lookup then costs ONE byte out of a
    // table of sigma^k bytes (L2 / Infinity-Cache resident) and its hit list ONE aligned line of the arena, where the
    // contiguous copy costs two table entries out of a 4-byte-per-key table and a run that straddles lines 1.3 times on average.
    // Derived from offs and the contiguous copy when the index is installed; not part of the image.  NULL: no cells.
    const uint8_t* cnt8;
    uint64_t cell_base;    // arena index of cell 0 (a multiple of 32)
    uint32_t cell_shift;   // log2 of the cell size in positions: 3, 4 or 5 (32-, 64- or 128-byte cells)
    // PREFIX LEVELS (dense elements; derived when the index is installed, not part of the image): level L = 1 .. n_levels holds,
    // for every (k - L)-mer in rank-hash order, the ascending list of ALL its occurrences — what get_position_for_all_kmer_with_prefix
    // + check_last_kmer + to_vector() (kmer_index.hpp:90-148, kmer_index_result.hpp:244-260) return for it, i.e. the sigma^L
    // buckets merged once, ahead of time, instead of per query.  A query of m < k letters is answered from level
    // min(k - m, n_levels): from ONE list when the level has its length, otherwise from sigma^(k - L - m) lists instead of
    // sigma^(k - m) buckets.  lvl_offs_at[L - 1]: arena index of the level's sigma^(k-L) + 1 list boundaries (relative to
    // lvl_base[L - 1], the arena index of its n - (k - L) + 1 positions).
    uint32_t n_levels;
    uint64_t lvl_offs_at[KMX_MAX_LEVELS];
    uint64_t lvl_base[KMX_MAX_LEVELS];
};

// Planner entry for one query length m — what kmer_index::search consults at
// kmer_index.hpp:512-518 (_use_multi_search_scheme[m], _optimal_nk_sum[m]).
//   SINGLE: serve the whole query from element `elem` (:512-513)
//   MULTI : `elem` is the element of the LAST summand of _optimal_nk_sum[m]; the
//           summands before it are plan[m - k_last] (the DP builds list(q) =
//           list(q-k) + [k], :434-435); nparts = number of summands.
//   REPLANNED (engine planner table only): a MULTI entry of the reference (elem / nparts as above, so that the chains of longer
//           sums still walk through it) that the ENGINE answers as SINGLE on element nparts >> KMX_PLAN_ALT_SHIFT — the largest k
//           of the index: a sum of two 10-mers is also two overlapping 12-mers, whose buckets are sigma^2 times shorter.
enum { KMX_SCHEME_NONE = 0, KMX_SCHEME_SINGLE = 1, KMX_SCHEME_MULTI = 2, KMX_SCHEME_REPLANNED = 3 };
#define KMX_PLAN_ALT_SHIFT 11                                   // nparts < 2^11 (range <= 65535 * 9, summands >= 9 letters ... see make_fast_plan_entries)
#define KMX_PLAN_NPARTS_MASK ((1u << KMX_PLAN_ALT_SHIFT) - 1u)
struct KmxPlanEntry {
    uint8_t scheme;
    uint8_t elem;
    uint16_t nparts;
};

struct KmxIndexDev {
    uint64_t n;            // text length
    uint32_t sigma;
    uint32_t n_ks;
    uint32_t kmax;
    uint32_t range;        // query lengths >= range are rejected (kmer_index.hpp:507-509)
    const uint32_t* arena; // every element's positions, back to back
    const uint8_t* tail;   // last kmax letters of the text (the _last_kmer of every element, :174)
    const KmxPlanEntry* plan; // [range]
    uint64_t arena_elems;  // number of positions in the arena
    unsigned long long* dbg; // KMX_CHECKED builds: 16 words of violation records
    uint64_t pw[64];       // sigma^j (fast_pow(sigma, j)), saturated to ~0 on overflow
    KmxElemDev elems[KMX_MAX_KS];
};

// PREFIX queries whose slice has at most KMX_PSORT_CAP positions are sorted in LDS by one wave (k_prefix_sort_small:
// merges by rank up to 4 runs, a bitonic sort beyond — the number of runs does not matter to that one: a protein index
// has 20 runs of three positions behind every (k-1)-letter query); larger ones go through the global merge passes.
#define KMX_PSORT_IS_SMALL(runs, len) ((len) <= KMX_PSORT_CAP)
// ... of those, the slices merged by k_prefix_merge_small: more than the register paths of k_prefix_sort_small take (four runs,
// 512 positions), at most KMX_PSORT_MAX_RUNS runs, and runs long enough on average to be worth a round's bookkeeping
#ifndef KMX_PMERGE_MIN_AVG
#define KMX_PMERGE_MIN_AVG 8
#endif
#define KMX_PSORT_IS_MERGE(runs, len)                                                                                   \
    ((len) <= KMX_PSORT_CAP && (runs) >= 2 && (runs) <= KMX_PSORT_MAX_RUNS && !((runs) <= KMX_PMERGE_REG_RUNS && (len) <= KMX_PMERGE_REG_LEN) && \
     (len) >= KMX_PMERGE_MIN_AVG * (runs))
#ifndef KMX_PMERGE_REG_RUNS
#define KMX_PMERGE_REG_RUNS 4
#endif
#ifndef KMX_PMERGE_REG_LEN
#define KMX_PMERGE_REG_LEN 512
#endif
#define KMX_VRESOLVE 4          // k_lookup follows up to this many candidates of a single-k query through its parts itself
// Bytes allocated past the last arena element.  Kernels read 16 bytes at any element (k_fill, the staging of k_validate), and
// k_validate reads a chunk of candidates through one pointer clamped to the bucket's last entry + immediate offsets of up to
// 7 rounds x 16 lanes x 4 bytes: both stay inside the allocation whatever the bucket.
// ... and k_fill sends the slots it must not write (a PREFIX slice the sort kernels produce themselves) to read the padding,
// which holds 0xFFFFFFFF ("do not store") for a whole tile of slots: 4096 words + the 1024 bytes above.
#define KMX_ARENA_PAD (4096 * 4 + 1024)
#define KMX_VTINY 8             // k_validate_tiny: one thread per STITCH query up to this many candidates / filter entries
#define KMX_VSHORT 16           // k_validate_short: two-part STITCH queries whose SHORTER bucket has at most this many entries (one per lane of a
                                // 16-lane group) and whose longer one fits a group's stage — the shorter side is walked, whichever part it is
#define KMX_PSORT_MAX_RUNS 32   // run boundaries kept per query by k_prefix_sort_small (its merge paths need 4 of them)
#define KMX_PSORT_CAP 2048
// ... up to KMX_PSORT_BLOCK_CAP positions (any number of runs) by one 1024-thread block (bitonic sort in
// 128 KB of LDS, k_prefix_sort_block); beyond that the global merge passes.
#define KMX_PSORT_BLOCK_CAP 32768
// of those, slices up to KMX_PSORT_MID_CAP positions take a 256-thread block with a quarter of the LDS (four per CU)
#define KMX_PSORT_MID_CAP 8192

// k_small — the latency path of small batches (kmer_index::search(query) is a batch of one): ONE launch of up to
// KMX_SMALL_BLOCKS workgroups, 256 queries each, that read the queries from and write the complete result to ONE page-locked
// host block ("mailbox"): per workgroup an input area, then the result arrays of the whole batch, which the workgroups fill
// at offsets they agree on among themselves (hit and mask-word totals exchanged through device memory).
#define KMX_SMALL_NQ 256          // queries per workgroup
#define KMX_SMALL_BLOCKS 32       // workgroups per launch: batches up to 8192 queries take the latency path
struct KmxSmallArgs {             // per workgroup: its number of queries and of letters (by value in the kernel arguments)
    uint16_t nq[KMX_SMALL_BLOCKS];
    uint16_t n_letters[KMX_SMALL_BLOCKS];
};
#define KMX_SMALL_IN_BYTES 8192   // per workgroup: (nq + 1) offsets + letters, staged in LDS
// "slow" queries — cross-referenced ones (STITCH) and sub-k ones whose slice has several runs (PREFIX, wants sorting):
// up to KMX_SMALL_WSLOW of them with at most KMX_SMALL_WCAP candidates / positions are taken by the workgroup's four waves in
// parallel, up to KMX_SMALL_BSLOW bigger ones (at most KMX_SMALL_SORT) by the whole workgroup one after the other
#define KMX_SMALL_WCAP 1024
#define KMX_SMALL_WSLOW 32
#define KMX_SMALL_SORT 4096
#define KMX_SMALL_BSLOW 8
#define KMX_SMALL_WORDS (KMX_SMALL_WSLOW * (KMX_SMALL_WCAP / 64 + 1) + KMX_SMALL_BSLOW * (KMX_SMALL_SORT / 64 + 1))   // mask words per workgroup
#define KMX_SMALL_POS 49152       // hit positions per workgroup
struct KmxSmallHeader {           // one per workgroup
    uint32_t fallback;            // != 0: the batch is not for this kernel (too many hits / slow queries, or an exchange that ran out): nothing else is valid
    uint32_t nq;
    uint64_t n_hits, n_mask_words;
    uint32_t n_stitch, n_prefix, n_error, n_none;
    uint32_t pad[6];
};
// byte offsets of a mailbox laid out for `blocks` workgroups:
//   [blocks x input area][blocks x KmxSmallHeader][hit_off (256 blocks + 1) u64][cand_src u64][mask_base u64][cand_count u32]
//   [status u8][kinds u8][mask words][positions]
struct KmxSmallLayout {
    uint32_t blocks, off_header, off_hitoff, off_csrc, off_mbase, off_ccnt, off_status, off_kinds, off_words, off_pos, bytes, pad;
};
static inline KmxSmallLayout kmx_small_layout(uint32_t blocks)
{
    KmxSmallLayout L;
    const uint32_t nq = blocks * KMX_SMALL_NQ;
    L.blocks = blocks;
    L.off_header = blocks * KMX_SMALL_IN_BYTES;
    L.off_hitoff = L.off_header + blocks * 64;
    L.off_csrc = L.off_hitoff + (nq + 2) * 8;
    L.off_mbase = L.off_csrc + nq * 8;
    L.off_ccnt = L.off_mbase + nq * 8;
    L.off_status = L.off_ccnt + nq * 4;
    L.off_kinds = L.off_status + nq;
    L.off_words = L.off_kinds + nq;
    L.off_pos = L.off_words + blocks * KMX_SMALL_WORDS * 8;
    L.bytes = L.off_pos + blocks * KMX_SMALL_POS * 4;
    L.pad = 0;
    return L;
}

// Counter block written by the lookup kernel and read back once per batch.
enum {
    KMX_CTR_STITCH = 1,
    KMX_CTR_PREFIX = 2,
    KMX_CTR_ERROR = 3,
    KMX_CTR_MASK_WORDS = 4,   // bump allocator for STITCH mask words
    KMX_CTR_PREFIX_ELEMS = 5, // sum of the slice lengths of the LARGE PREFIX queries (global merge passes)
    KMX_CTR_MAX_RUNS = 6,     // max number of runs of any LARGE PREFIX query
    KMX_CTR_TOTAL_HITS = 7,   // written by the scan
    KMX_CTR_NONE = 8,         // valid queries without a hit
    KMX_CTR_PREFIX_TOTAL = 9, // total of the scan over PREFIX slice lengths
    KMX_CTR_STITCH_MORE = 10, // STITCH queries with more further parts than the one QueryDesc::p1 names
    KMX_CTR_PREFIX_BIG = 11,  // PREFIX queries that are not 'small' (listed from the BACK of prefix_list)
    KMX_CTR_STITCH_TINY = 12, // STITCH queries with at most KMX_VTINY candidates and filter-bucket entries (listed from the BACK of stitch_list)
    KMX_CTR_STITCH_RESOLVED = 13, // STITCH queries k_lookup resolved by itself (tiny first bucket, survivors one run of it)
    KMX_CTR_PREFIX_MERGE = 14, // small PREFIX queries of the merge class (KMX_PSORT_IS_MERGE): k_prefix_merge_small has work
    KMX_CTR_PREFIX_MID = 15,   // of KMX_CTR_PREFIX_BIG: slices of at most KMX_PSORT_MID_CAP positions (k_prefix_sort_block's 256-thread variant)
    KMX_CTR_PREFIX_PLAIN = 0,  // PREFIX queries answered by ONE list (nothing to sort: on no work list)
    KMX_CTR_STITCH_SHORT = 16, // two-part STITCH queries whose shorter bucket has at most KMX_VSHORT entries (listed in QueryDesc::short_list)
    KMX_CTR_LONG = 17,         // single-k queries of more than KMX_LONG_PARTS parts (k_lookup_long takes them when the batch before had some)
    KMX_CTR_PSB_MERGE = 18,    // chunks of the longer PREFIX slices listed for k_prefix_merge_block (k_prefix_items counts; KMX_CTR_PSB_OTHER follows it)
    KMX_CTR_PSB_OTHER = 19,    // ... and for k_prefix_sort_items (more runs than the merge takes)
    KMX_CTR_PSB_BANDS = 20,    // bands of the PREFIX slices k_prefix_bands cut (k_prefix_merge_band's work; KMX_CTR_PSB_CUTS follows it)
    KMX_CTR_PSB_CUTS = 21,     // words of cut tables handed out
    KMX_CTR_PSB_SPLITS = 22,   // slices spread by value (k_prefix_split_*), then: words of their counters (23), of their scratch space (24), their tiles (25)
    KMX_CTR_COUNT = 28
};
#define KMX_LONG_PARTS 256      // parts beyond which a query's probes are spread over the lanes of a wave instead of walked by one lane
                                // (measured: 100 parts — 1000 letters on k = 10 — are faster walked, 0.76 against 2.5 ms per 1e5 reads: a
                                //  wave per query pays its serial epilogue per query; 500 parts 0.56 against 3.76 ms per 2e4 reads)
#define KMX_SEARCH_INTERNAL_DEFER_LONG 0x80000000u   // (k_lookup's flags, set by the engine) such queries are listed for k_lookup_long, not walked
