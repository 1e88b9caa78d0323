// Hand-written gfx950 (CDNA4, wave64) kernels of the batch search path.
//
// Pipeline for one batch (launch order; [..] only when such queries exist):
//   k_lookup        one query per lane: rank-hash (kmer_index.hpp:56-73), planner
//                   lookup (:512-518), bucket probe (:76-84), sub-k prefix range +
//                   last-kmer fix-up (:115-148) -> per-query descriptor
//  [k_validate]     four STITCH queries per wave: the candidates of the first part against one further
//                   part's bucket staged in LDS (:270-298, :532-555) -> compressed_bitset mask words via
//                   ballot + the compacted survivors
//  [k_validate_more] queries with more parts: the survivors against every part, one part per lane
//  [k_validate_tiny] STITCH queries with a handful of candidates (large k): one thread per query does all of it
//  [k_validate_wave] long survivor lists and big queries (repeats of the text): one wave per query, smallest bucket as anchor
//   k_scan_*        exclusive scan of the per-query hit counts -> hit_off
//   k_partition     first query of every output tile
//   k_fill          output-centric copy of bucket runs -> to_vector() lists
//  [k_compact]      STITCH fallback without a survivor buffer: mask-word decode, popcount prefix compaction
//  [k_prefix_*]     PREFIX slices of several runs: merged into one ascending list (the std::sort of
//                   kmer_index_result.hpp:258); slices of one run (prefix levels) need none of them.  By slice length and runs:
//                     <= 2048 positions        a wave per slice: k_prefix_sort_small / k_prefix_merge_small
//                     <= 8192                  a 256-thread block per slice: k_prefix_sort_block
//                     <= 32768 (one chunk)     records from k_prefix_items; k_prefix_merge_block (<= 128 runs: merge rounds in LDS, the
//                                              next chunk's loads and the previous chunk's stores under them) or k_prefix_sort_items
//                                              (more runs: distribution sort)
//                     beyond, <= 64 runs       cut by value into bands (k_prefix_bands), a 256-thread block per band (k_prefix_merge_band)
//                     beyond, more runs        spread by value (k_prefix_split_count / _scan / _scatter), a band per k_prefix_sort_items block
//                     beyond, positions that crowd   chunks + k_prefix_merge_pass, ceil(log2 chunks) times
//  [k_validate_wide / k_validate_more_thread]  long filter buckets (linear intersection) / few further parts per query
//
// All arithmetic is unsigned integer; no MFMA.  The kernels are HBM / latency
// bound: see DESIGN.md for bytes per unit and the roofline of each.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <type_traits>

#include "kmx_kernels.h"

#define KMX_BLOCK 256
#define KMX_WAVE 64
#define KMX_PSORT_PAIR_CAP 512   // k_prefix_sort_small: slices up to this length are merged pairwise from registers
#ifndef KMX_PSORT_MULTIWAY_RUNS
#define KMX_PSORT_MULTIWAY_RUNS 4   // k_prefix_sort_small: multi-way rank pass up to this many runs, bitonic beyond
#endif
// merge_runs_lds: steps per thread and LDS slack of the wave-level (k_prefix_merge_small) and the block-level user
#define KMX_PM_WAVE_EMAX (((KMX_PSORT_CAP + (KMX_WAVE - KMX_PSORT_MAX_RUNS / 2) - 1) / (KMX_WAVE - KMX_PSORT_MAX_RUNS / 2)) | 1)   // 43
#define KMX_PM_WAVE_PAD (KMX_PSORT_MAX_RUNS + KMX_PM_WAVE_EMAX + 5)   // sentinel cells + the reads of a chunk past its end
#define KMX_PM_TILE 4096     // k_prefix_merge_pass: output positions per workgroup (divides KMX_PSORT_BLOCK_CAP)
#define KMX_STAGE_CAP 1024   // k_validate: part-bucket entries staged in LDS per wave
// k_validate: KMX_VGROUPS queries per wave, one per KMX_VGROUP-lane group
#define KMX_VGROUP 16                            // 32 and 64 lanes per query measured slower
#define KMX_VGROUPS (KMX_WAVE / KMX_VGROUP)
#define KMX_VSLICES (64 / KMX_VGROUP)            // ballot slices per mask word
#define KMX_VSTAGE (1024 / KMX_VGROUPS)          // staged bucket entries per group (4 KB of LDS per wave)
#define KMX_VCH 8                                // candidates per lane and chunk, searched in lockstep

namespace kmx {

// Flag in QueryDesc::src: the query is not a plain bucket copy (STITCH / PREFIX): k_fill
// serves PREFIX slot by slot and leaves STITCH to k_compact.
#define SRC_SLOW (uint64_t(1) << 63)
// additionally set for PREFIX queries: k_fill copies their slice on its fast path, only the (rare)
// last-kmer positions behind the slice take the per-slot path
#define SRC_PREFIX (uint64_t(1) << 62)
#define SRC_FLAGS (SRC_SLOW | SRC_PREFIX)
#define KMX_P1_MORE (uint64_t(1) << 63)    // QueryDesc::p1: the query has further parts beyond the one p1 names
#define KMX_P1_BIG (uint64_t(1) << 62)     // the first part's bucket is long (a repeat of the text): k_validate leaves the
                                           // query to validate_big_wave, which anchors it on its smallest bucket
#define KMX_P1_SWAP (uint64_t(1) << 61)    // short class only (k_validate_short): the part p1 names has the SHORTER bucket — its entries
                                           // are walked and looked up in the first part's bucket instead of the other way round
#define KMX_P1_DELTA_MASK 0x1FFFFFFFu      // offset of the filter part in the query (bits 32..60 of p1)
#define KMX_VBIG 1024                      // candidates beyond which a STITCH query counts as big
// A filter bucket of more than 256 entries does not fit the stage of a 16-lane group in k_validate; up to KMX_VWIDE entries
// the query gets a wave of its own with the whole stage (k_validate_wide) instead of searching the bucket where it lies.
#define KMX_VWIDE_MIN 256
#define KMX_VWIDE 2048

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
// KMX_P1_BIG: the first part's bucket is long AND anchoring the query on its smallest bucket pays — the filter bucket is
// much shorter than the candidates, or too long for any stage.  Two long buckets of the same order (a small k over a long
// text: 13-letter reads on a k = 8 element, 1526 positions per bucket at 1e8 letters) are the wide path's case.
__device__ __forceinline__ bool stitch_goes_big(uint32_t first_cnt, uint32_t filter_cnt)
{
    return first_cnt > KMX_VBIG && (filter_cnt > KMX_VWIDE || uint64_t(filter_cnt) * 8 < first_cnt);
}
__device__ __forceinline__ bool stitch_is_wide(uint64_t p1)
{
    return !(p1 & KMX_P1_BIG) && uint32_t(p1) > KMX_VWIDE_MIN && uint32_t(p1) <= KMX_VWIDE;
}
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (KMX_WAVE - 1); }

// Pointers that are loaded from the index header are "generic" to the compiler, which then
// emits flat_load (counted on BOTH vmcnt and lgkmcnt, so every LDS wait would also drain the
// outstanding global loads).  Everything the header points to lives in HBM: say so.
#define KMX_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const KMX_GLOBAL T* as_global(const T* p)
{
    return (const KMX_GLOBAL T*)p;
}
#define KMX_LDS __attribute__((address_space(3)))      // a pointer that is KNOWN to point into LDS (ds_read, not flat_load)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
// a quad of sorted positions on its way out (the sub-k kernels' copy-outs): streamed, nothing on the device reads it again soon
#ifdef KMX_NT_QUAD_LOADS
#define KMX_LOAD_QUAD(p) __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4*>(p))
#else
#define KMX_LOAD_QUAD(p) (*reinterpret_cast<const u32x4_a4*>(p))
#endif
#ifdef KMX_PLAIN_QUAD_STORES
#define KMX_STORE_QUAD(p, q) (*reinterpret_cast<u32x4_a4*>(p) = (q))
#else
#define KMX_STORE_QUAD(p, q) __builtin_nontemporal_store((q), reinterpret_cast<u32x4_a4*>(p))
#endif

// plan[m] as one 32-bit load ({u8 scheme, u8 elem, u16 nparts}, little endian).
// plan_effective: the entry a query of m letters is ANSWERED by — a KMX_SCHEME_REPLANNED entry (engine planner table) reads
// as SINGLE on its alternative element; plan_chain: the entry as the reference's DP wrote it (elem = last summand), which is
// what a walk over the summands of a longer sum wants (:434-435).
__device__ __forceinline__ uint32_t plan_effective(uint32_t raw)
{
    return (raw & 0xFF) == KMX_SCHEME_REPLANNED ? (uint32_t(KMX_SCHEME_SINGLE) | ((raw >> (16 + KMX_PLAN_ALT_SHIFT)) << 8) | (1u << 16)) : raw;
}
__device__ __forceinline__ KmxPlanEntry plan_unpack(uint32_t raw)
{
    KmxPlanEntry e;
    e.scheme = uint8_t(raw & 0xFF);
    e.elem = uint8_t((raw >> 8) & 0xFF);
    e.nparts = uint16_t((raw >> 16) & KMX_PLAN_NPARTS_MASK);
    return e;
}
__device__ __forceinline__ KmxPlanEntry load_plan(const KmxIndexDev* __restrict__ ix, uint64_t m)
{
    return plan_unpack(plan_effective(((const KMX_GLOBAL uint32_t*)ix->plan)[m]));
}
__device__ __forceinline__ KmxPlanEntry load_plan_chain(const KmxIndexDev* __restrict__ ix, uint64_t m)
{
    return plan_unpack(((const KMX_GLOBAL uint32_t*)ix->plan)[m]);
}

struct Run {
    uint64_t src;   // arena index of the run's first position
    uint32_t cnt;
};

__device__ __forceinline__ uint64_t slot_hash_dev(uint64_t key, uint32_t log2cap)
{
    return log2cap ? (key * 0x9E3779B97F4A7C15ull) >> (64 - log2cap) : 0;
}

// size of a group from two consecutive packed entries of KmxElemDev::atab
__device__ __forceinline__ uint32_t atab_count(uint32_t e0, uint32_t e1)
{
    const uint32_t padded = (e1 & ~31u) - (e0 & ~31u), r = e0 & 31u;
    return padded ? (r ? padded - 32u + r : padded) : 0u;
}

// at(hash) — kmer_index.hpp:76-84: the bucket of one rank-hash, or cnt == 0.
__device__ __forceinline__ Run probe(const KmxElemDev* __restrict__ el, uint64_t h)
{
    Run r;
    if (el->table_kind == KMX_TABLE_DENSE) {
        if (el->cnt8) {                                       // short buckets: one byte says how much of the key's cell is in use
            const uint32_t c = as_global(el->cnt8)[h];
            if (c != 255u) { r.src = el->cell_base + (h << el->cell_shift); r.cnt = c; return r; }
        }
        if (el->atab) {                                       // the line-aligned copy of the bucket
            const KMX_GLOBAL uint32_t* atab = as_global(el->atab);
            const uint32_t e0 = atab[h], e1 = atab[h + 1];
            r.src = el->arena_base + (e0 & ~31u);
            r.cnt = atab_count(e0, e1);
            return r;
        }
        const KMX_GLOBAL uint32_t* offs = as_global(el->offs);
        uint32_t a = offs[h], b = offs[h + 1];
        r.cnt = b - a;
        r.src = el->arena_base + a;
        return r;
    }
    const KMX_GLOBAL KmxSlot* slots = as_global(el->slots);
    const uint64_t mask = (uint64_t(1) << el->log2cap) - 1;
    uint64_t s = slot_hash_dev(h, el->log2cap);
    for (;;) {
        // one 16-byte slot: {key, off, cnt}
        const u64x2 raw = *(const KMX_GLOBAL u64x2*)(slots + s);
        uint32_t off = uint32_t(raw.y), cnt = uint32_t(raw.y >> 32);
        if (cnt == 0) { r.src = 0; r.cnt = 0; return r; }
        if (raw.x == h) { r.src = el->arena_base + off; r.cnt = cnt; return r; }
        s = (s + 1) & mask;
    }
}

// first index i in [0, n) with a[i] >= x (n if none)
template <typename T, typename P>
__device__ __forceinline__ uint64_t lower_bound_dev(P a, uint64_t n, T x)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// first index i in [0, n) with a[i] > x (n if none)
template <typename T, typename P>
__device__ __forceinline__ uint64_t upper_bound_dev(P a, uint64_t n, T x)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (a[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The run boundaries of a PREFIX query of m letters whose first key is `key` (QueryDesc::key): c0 + 1 entries of the offset
// table lookup_query took the slice from — the element's, or its prefix level's.
__device__ __forceinline__ const KMX_GLOBAL uint32_t* prefix_run_bounds(const KmxIndexDev* __restrict__ ix, uint64_t m, uint64_t key)
{
    const KmxPlanEntry pe = load_plan(ix, m);
    const KmxElemDev* el = &ix->elems[pe.elem];
    const uint32_t L = min(uint32_t(el->k - m), el->n_levels);
    return (L ? as_global(ix->arena) + el->lvl_offs_at[L - 1] : as_global(el->offs)) + key;
}

// rank-hash of `len` letters by Horner's rule; equals sum r_i * sigma^(len-i-1)
// (kmer_index.hpp:56-73) because k < 64/log2(sigma) rules out wrap-around.
// Returns false when a letter is not a valid rank.
// The letters are fetched 16 at a time with one byte-aligned 16-byte load (instead of one load
// per letter) whenever those 16 bytes lie inside the query buffer [.., qend).
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ bool rank_hash(const uint8_t* __restrict__ q, uint32_t len, uint32_t sigma, uint64_t& h,
                                          const uint8_t* __restrict__ qend)
{
    uint64_t acc = 0;
    bool ok = true;
    for (uint32_t c = 0; c < len; c += 16) {
        const uint32_t nb = min(16u, len - c);
        if (q + c + 16 <= qend) {
            const u32x4_a1 w = *reinterpret_cast<const u32x4_a1*>(q + c);
            uint64_t lo = uint64_t(w[0]) | (uint64_t(w[1]) << 32), hi = uint64_t(w[2]) | (uint64_t(w[3]) << 32);
            for (uint32_t j = 0; j < nb; ++j) {
                const uint32_t r = uint32_t(lo & 0xFF);
                lo = (lo >> 8) | (hi << 56);
                hi >>= 8;
                ok &= r < sigma;
                acc = acc * sigma + r;
            }
        } else {
            for (uint32_t j = 0; j < nb; ++j) {
                const uint32_t r = q[c + j];
                ok &= r < sigma;
                acc = acc * sigma + r;
            }
        }
    }
    h = acc;
    return ok;
}

// bitonic sort of n2 (power of two) LDS words by `nthreads` cooperating threads; `sync` separates the stages
template <typename Sync>
__device__ __forceinline__ void bitonic_lds(uint32_t* sbuf, uint32_t n2, uint32_t tid, uint32_t nthreads, Sync sync)
{
    for (uint32_t size = 2; size <= n2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < n2 / 2; t += nthreads) {
                const uint32_t lo = 2 * t - (t & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint32_t a = sbuf[lo], b = sbuf[hi];
                if ((a > b) == up) { sbuf[lo] = b; sbuf[hi] = a; }
            }
            sync();
        }
    }
}

// The per-query half of kmer_index::search (kmer_index.hpp:505-558 -> :193-346): planner entry, rank-hashes, bucket probes,
// the sub-k prefix range with its last-kmer fix-up, the parts of a long query.  Shared by k_lookup (every query that is
// not a plain exact lookup) and k_small (all of them).  elems_s: the element descriptors (LDS copy or ix->elems).
struct LookupOut {
    uint64_t src, aux, key, p1;   // as QueryDesc
    uint32_t cnt, c0;
    uint8_t kind, status;
    bool resolved;                // a STITCH query whose few candidates were followed through every part right here
};
__device__ __forceinline__ void lookup_query(const KmxIndexDev* __restrict__ ix, const KmxElemDev* elems_s,
                                             const uint8_t* __restrict__ qr, uint64_t m, const uint8_t* __restrict__ qend,
                                             uint32_t flags, LookupOut& o)
{
    uint8_t kind = KMX_KIND_NONE, status = KMX_Q_OK;
    uint64_t src = 0, aux = 0, key = 0;
    uint64_t p1 = 0;              // STITCH: one further part as (offset in query << 32) | bucket size, bucket in `key`; bit 63: more parts follow
    uint32_t cnt = 0, c0 = 0;
    bool resolved = false;
    const uint32_t sigma = ix->sigma;
    if (m == 0) {
        status = KMX_Q_EMPTY_QUERY;                       // assert(query.size() > 0), kmer_index.hpp:195
    } else if (m >= ix->range) {
        status = KMX_Q_TOO_LONG;                          // :507-509
    } else {
        const KmxPlanEntry pe = load_plan(ix, m);
        bool ranks_ok = true;
        // Large k: the buckets hold a handful of positions.  Up to KMX_VRESOLVE start positions of the query are
        // followed through the parts while those are probed anyway (p + offset_j must be in part j's bucket,
        // :279-291, :541-551); when the survivors are one run of the first part's bucket the query needs no
        // validation pass, no mask words and no second read-back: it is filled like an exact lookup.  Not with
        // KEEP_MASKS (the words are the point there).
        // Only for elements whose buckets average at most two positions: with longer buckets the extra loads cost more
        // than the validation they save (BASELINE config 3: +0.16 ms of lookup for -0.05 ms of validation).
        auto sparse_buckets = [](const KmxElemDev* e) {
            return e->npos <= 2 * (e->table_kind == KMX_TABLE_DENSE ? e->n_keys : uint64_t(e->n_ukeys));
        };
        const KMX_GLOBAL uint32_t* ar = as_global(ix->arena);
        uint32_t cand[KMX_VRESOLVE];
        uint32_t alive = 0;
        bool track = false;
        auto follow = [&](const Run& r, uint32_t delta) {
            if (r.cnt > KMX_VTINY) { track = false; return; }
            uint32_t found = 0;
            for (uint32_t t = 0; t < r.cnt; ++t) {
                const uint32_t b = ar[r.src + t];
#pragma unroll
                for (uint32_t i = 0; i < KMX_VRESOLVE; ++i) found |= uint32_t(cand[i] + delta == b) << i;
            }
            alive &= found;
        };
        if (pe.scheme == KMX_SCHEME_SINGLE) {
            const KmxElemDev* el = &elems_s[pe.elem];
            const uint32_t k = el->k;
            if (m == k) {                                 // :198-205
                uint64_t h;
                ranks_ok = rank_hash(qr, k, sigma, h, qend);
                if (ranks_ok) {
                    Run r = probe(el, h);
                    if (r.cnt) { kind = KMX_KIND_EXACT; src = r.src; cnt = r.cnt; }
                }
            } else if (m < k) {                           // :342-345 -> :115-148
                if (ix->pw[k - m] > KMX_SUBK_FANOUT_LIMIT) {   // fast_pow(sigma, k - size)
                    status = KMX_Q_SUBK_FANOUT;           // :119-122
                } else {
                    uint64_t hp;
                    ranks_ok = rank_hash(qr, uint32_t(m), sigma, hp, qend);
                    if (ranks_ok) {
                        // prefix levels (KmxElemDev::n_levels): level L lists every (k - L)-mer's occurrences; the query is
                        // answered at kk = k - L as if that were the element's k — its sigma^(kk - m) lists, its last-kmer letters
                        const uint32_t L = min(uint32_t(k - m), el->n_levels);
                        const uint32_t kk = k - L;
                        const uint64_t R = ix->pw[kk - m];
                        hp *= R;                          // prefix_hash, :124-129
                        uint64_t klo, khi;                // key-index range of the prefix
                        const KMX_GLOBAL uint32_t* run_offs = as_global(el->offs);
                        uint64_t run_base = el->arena_base;
                        if (L) {
                            klo = hp; khi = hp + R;
                            run_offs = as_global(ix->arena) + el->lvl_offs_at[L - 1];
                            run_base = el->lvl_base[L - 1];
                        } else if (el->table_kind == KMX_TABLE_DENSE) {
                            klo = hp; khi = hp + R;
                        } else {
                            const KMX_GLOBAL uint64_t* uk = as_global(el->ukeys);
                            auto first_key_at_least = [&](uint64_t v) -> uint64_t {
                                if (!el->dir) return lower_bound_dev<uint64_t>(uk, el->n_ukeys, v);
                                const uint64_t j = v >> el->dir_shift;              // the directory cell of v
                                if (j >= el->n_dir) return el->n_ukeys;
                                const uint32_t c_lo = as_global(el->dir)[j], c_hi = as_global(el->dir)[j + 1];
                                return c_lo + lower_bound_dev<uint64_t>(uk + c_lo, c_hi - c_lo, v);
                            };
                            klo = first_key_at_least(hp);
                            khi = first_key_at_least(hp + R);
                        }
                        const uint32_t lo = run_offs[klo], hi = run_offs[khi];
                        // check_last_kmer, :90-112: offsets n-k+i, i in [1, k-m], where no k-mer
                        // starts but the query still fits.  Bit j of aux <-> position n - j.
                        const KMX_GLOBAL uint8_t* tail = as_global(ix->tail) + (ix->kmax - kk);   // last kk letters
                        uint64_t tmask = 0;
                        for (uint32_t i = 1; i + m <= kk; ++i) {
                            bool eq = true;
                            for (uint32_t t = 0; t < m; ++t) eq &= tail[i + t] == qr[t];
                            if (eq) tmask |= uint64_t(1) << (kk - i);
                        }
                        const uint32_t len = hi - lo;
                        cnt = len + uint32_t(__popcll(tmask));
                        if (cnt) {
                            kind = KMX_KIND_PREFIX;
                            src = run_base + lo;
                            aux = tmask;
                            key = klo;
                            c0 = uint32_t(khi - klo);     // number of runs
                                                    }
                    }
                }
            } else {                                      // m > k, :207-339
                const uint32_t P = uint32_t(m / k), rest = uint32_t(m % k);
                bool all = true;
                Run first{0, 0}, extra{0, 0};
                uint32_t extra_delta = 0;
                for (uint32_t j = 0; j < P && all && ranks_ok; ++j) {   // :216-227
                    uint64_t h;
                    ranks_ok = rank_hash(qr + uint64_t(j) * k, k, sigma, h, qend);
                    if (!ranks_ok) break;
                    Run r = probe(el, h);
                    if (j == 0) {
                        first = r;
                        track = !(flags & KMX_SEARCH_KEEP_MASKS) && sparse_buckets(el) && r.cnt != 0 && r.cnt <= KMX_VRESOLVE;
                        if (track) {
#pragma unroll
                            for (uint32_t i = 0; i < KMX_VRESOLVE; ++i) cand[i] = ar[r.src + (i < r.cnt ? i : 0u)];
                            alive = (1u << r.cnt) - 1u;
                        }
                    } else {
                        extra = r; extra_delta = j * k;
                        if (track && r.cnt) follow(r, j * k);
                    }
                    all = r.cnt != 0;
                }
                if (all && ranks_ok) {
                    if (rest && ix->pw[k - rest] > KMX_SUBK_FANOUT_LIMIT) {
                        status = KMX_Q_SUBK_FANOUT;       // the rest lookup throws at :119-122 via :234
                    } else {
                        if (rest) {
                            // the rest is verified through the k-mer that ENDS the query
                            uint64_t h;
                            ranks_ok = rank_hash(qr + (m - k), k, sigma, h, qend);
                            if (ranks_ok) { extra = probe(el, h); extra_delta = uint32_t(m - k); }
                            all = ranks_ok && extra.cnt != 0;
                            if (all && track) follow(extra, uint32_t(m - k));
                        }
                        if (all && ranks_ok) {
                            kind = KMX_KIND_STITCH; src = first.src; c0 = first.cnt;
                            key = extra.src;
                            p1 = (uint64_t(extra_delta) << 32) | extra.cnt | (P - 1 + (rest ? 1 : 0) > 1 ? KMX_P1_MORE : 0);
                            if (!(flags & KMX_SEARCH_KEEP_MASKS) && stitch_goes_big(first.cnt, extra.cnt)) p1 |= KMX_P1_BIG;
                            if (track) {
                                const uint32_t lo = alive ? uint32_t(__ffs(int(alive))) - 1u : 0u, len = uint32_t(__popc(alive));
                                if ((alive >> lo) == (1u << len) - 1u) {       // one run of the bucket (or nothing)
                                    resolved = true;
                                    src = first.src + lo;
                                    cnt = len;
                                }
                            }
                        }
                    }
                }
            }
        } else {                                          // multi-k scheme, :515-557
            // walk _optimal_nk_sum[m] from its last summand to its first
            uint64_t mm = m;
            bool all = true;
            Run r{0, 0}, extra{0, 0};
            uint32_t extra_delta = 0;
            const uint32_t nparts = pe.nparts;
            for (uint32_t j = 0; j < nparts && all && ranks_ok; ++j) {
                const KmxPlanEntry e = load_plan_chain(ix, mm);
                const KmxElemDev* el = &elems_s[e.elem];
                const uint32_t k = el->k;
                mm -= k;                                  // this summand covers [mm, mm + k)
                uint64_t h;
                ranks_ok = rank_hash(qr + mm, k, sigma, h, qend);
                if (!ranks_ok) break;
                r = probe(el, h);                         // search_k, :183-190 / :520
                if (j == 0) {
                    extra = r; extra_delta = uint32_t(mm);
                    // the walk meets the LAST summand first: its bucket anchors the start positions p = a - offset
                    track = !(flags & KMX_SEARCH_KEEP_MASKS) && nparts > 1 && sparse_buckets(el) && r.cnt != 0 && r.cnt <= KMX_VRESOLVE;
                    if (track) {
                        alive = 0;
#pragma unroll
                        for (uint32_t i = 0; i < KMX_VRESOLVE; ++i) {
                            const uint32_t a = ar[r.src + (i < r.cnt ? i : 0u)];
                            cand[i] = a - uint32_t(mm);
                            alive |= uint32_t(i < r.cnt && a >= uint32_t(mm)) << i;
                        }
                    }
                } else if (track && r.cnt) {
                    follow(r, uint32_t(mm));
                }
                all = r.cnt != 0;                         // :521-524
            }
            if (all && ranks_ok) {
                if (nparts == 1) { kind = KMX_KIND_EXACT; src = r.src; cnt = r.cnt; }   // :529-530
                else {
                    kind = KMX_KIND_STITCH; src = r.src; c0 = r.cnt;
                    key = extra.src;
                    p1 = (uint64_t(extra_delta) << 32) | extra.cnt | (nparts > 2 ? KMX_P1_MORE : 0);
                    if (!(flags & KMX_SEARCH_KEEP_MASKS) && stitch_goes_big(r.cnt, extra.cnt)) p1 |= KMX_P1_BIG;
                    if (track) {
                        // r is the first summand's bucket now (offset 0): which of its entries are surviving starts?
                        uint32_t fmask = 0;
                        for (uint32_t t = 0; t < r.cnt; ++t) {
                            const uint32_t b = ar[r.src + t];
#pragma unroll
                            for (uint32_t i = 0; i < KMX_VRESOLVE; ++i) fmask |= uint32_t(((alive >> i) & 1u) && cand[i] == b) << t;
                        }
                        const uint32_t lo = fmask ? uint32_t(__ffs(int(fmask))) - 1u : 0u, len = uint32_t(__popc(fmask));
                        if ((fmask >> lo) == (1u << len) - 1u) {           // one run of the bucket (or nothing)
                            resolved = true;
                            src = r.src + lo;
                            cnt = len;
                        }
                    }
                }
            }
        }
        if (!ranks_ok) { status = KMX_Q_BAD_RANK; kind = KMX_KIND_NONE; cnt = 0; }
    }
    o.src = src; o.aux = aux; o.key = key; o.p1 = p1; o.cnt = cnt; o.c0 = c0; o.kind = kind; o.status = status; o.resolved = resolved;
}

// ---------------------------------------------------------------------------
// k_lookup — one query per lane.
// ---------------------------------------------------------------------------
struct BlockCounters {
    unsigned int n_long, base_long;
    unsigned int n_stitch, n_stitch_tiny, n_stitch_short, n_resolved, n_prefix, n_prefix_big, n_prefix_merge, n_prefix_mid, n_prefix_plain, n_error, n_none, n_more;
    unsigned long long words, pelems, hits;
    unsigned int max_runs;
    unsigned int base_stitch, base_stitch_tiny, base_stitch_short, base_prefix, base_prefix_big;
    unsigned long long base_words;
    unsigned int max_fan_exp;       // largest j with sigma^j <= KMX_SUBK_FANOUT_LIMIT (the guard of kmer_index.hpp:119-122 as an exponent)
    unsigned int small_keys;        // every element's key space allows hash16<true>
};

// rank-hash (kmer_index.hpp:56-73) of the first `len` <= 16 letters of a 16-byte block that is already in registers.
// The letters are moved to the END of the block (a 128-bit shift by 16 - len bytes: the bytes in front of them read 0, and
// Horner's rule over leading zeros stays 0), so that the sixteen steps are straight-line code without a per-letter predicate: one
// byte extract, one compare and one multiply-add per letter.  SMALL (block-uniform): every partial sum fits 24 bits and the
// hash 32 — the multiply-add is the full-rate 24-bit one (a 32-bit integer multiply issues at a quarter of the rate).
template <bool SMALL>
__device__ __forceinline__ bool hash16(const u32x4_a1& w, uint32_t len, uint32_t sigma, uint64_t& h)
{
    uint64_t lo = uint64_t(w[0]) | (uint64_t(w[1]) << 32), hi = uint64_t(w[2]) | (uint64_t(w[3]) << 32);
    {
        const uint32_t sh = (16u - len) * 8u;                         // 0 .. 128 bits to the "left" (towards the last byte)
        const uint32_t s6 = sh & 63u;
        const uint64_t carry = s6 ? lo >> (64u - s6) : 0;
        const uint64_t hi1 = (hi << s6) | carry, lo1 = lo << s6;      // shift by sh mod 64 ...
        hi = sh >= 128u ? 0 : sh >= 64u ? lo1 : hi1;                  // ... and by whole 64-bit words
        lo = sh >= 64u ? 0 : lo1;
    }
    const uint32_t wd[4] = {uint32_t(lo), uint32_t(lo >> 32), uint32_t(hi), uint32_t(hi >> 32)};
    uint32_t top = 0;                                                 // the largest rank met: one compare at the end
    if (SMALL) {
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t r = (wd[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            top = max(top, r);
            acc = __umul24(acc, sigma) + r;
        }
        h = acc;
    } else {
        uint64_t acc = 0;
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t r = (wd[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            top = max(top, r);
            acc = acc * sigma + r;
        }
        h = acc;
    }
    return top < sigma;
}
// at(hash) (kmer_index.hpp:76-84) in two halves, so that the loads of several queries can be in flight together:
// where the first 16 bytes of the probe lie — dense: the table entries of h (cnt8 / atab / offs); open: the home slot —
__device__ __forceinline__ const char* probe_first_addr(const KmxElemDev* el, uint64_t h)
{
    if (el->table_kind == KMX_TABLE_DENSE)
        return el->cnt8 ? reinterpret_cast<const char*>(el->cnt8 + (h & ~uint64_t(15)))      // the aligned 16 bytes holding cnt8[h]
                        : reinterpret_cast<const char*>((el->atab ? el->atab : el->offs) + h);
    return reinterpret_cast<const char*>(el->slots + slot_hash_dev(h, el->log2cap));
}
// ... and the bucket they name (linear probing continues from the prefetched slot)
__device__ __forceinline__ Run probe_finish(const KmxElemDev* el, uint64_t h, const u32x4_a1& pr)
{
    Run r{0, 0};
    if (el->table_kind == KMX_TABLE_DENSE) {
        if (el->cnt8) {
            const uint32_t b = uint32_t(h) & 15u;
            const uint32_t wsel = (b >> 2) == 0 ? pr[0] : (b >> 2) == 1 ? pr[1] : (b >> 2) == 2 ? pr[2] : pr[3];
            const uint32_t c = (wsel >> ((b & 3u) * 8u)) & 0xFFu;
            if (c != 255u) { r.src = el->cell_base + (h << el->cell_shift); r.cnt = c; }
            else {                                    // (rare) the group is longer than a cell: the contiguous copy
                const KMX_GLOBAL uint32_t* offs = as_global(el->offs);
                const uint32_t a = offs[h], e = offs[h + 1];
                r.src = el->arena_base + a; r.cnt = e - a;
            }
        }
        else if (el->atab) { r.src = el->arena_base + (pr[0] & ~31u); r.cnt = atab_count(pr[0], pr[1]); }
        else { r.src = el->arena_base + pr[0]; r.cnt = pr[1] - pr[0]; }
        return r;
    }
    const KMX_GLOBAL KmxSlot* slots = as_global(el->slots);
    const uint64_t mask = (uint64_t(1) << el->log2cap) - 1;
    uint64_t s = slot_hash_dev(h, el->log2cap);
    uint64_t key = uint64_t(pr[0]) | (uint64_t(pr[1]) << 32);
    uint32_t off = pr[2], c = pr[3];
    while (c != 0 && key != h) {
        s = (s + 1) & mask;
        const u64x2 raw = *(const KMX_GLOBAL u64x2*)(slots + s);
        key = raw.x; off = uint32_t(raw.y); c = uint32_t(raw.y >> 32);
    }
    if (c) { r.src = el->arena_base + off; r.cnt = c; }
    return r;
}

#define KMX_PLAN_LDS 256        // planner entries kept in LDS by k_lookup (queries shorter than this take its interleaved passes)

#ifndef KMX_LOOKUP_OCC
#define KMX_LOOKUP_OCC 4        // waves per SIMD the two-part variant of k_lookup is compiled for (5 spills: measured slower)
#endif
// ITEMS: queries per thread (4; 8 for indexes of tiny cells, where a query is one byte of a table and one cell: twice the
// queries per wave keep more of those short round trips in flight).  PAIRS: the interleaved pass also takes the two-part
// cross-referenced queries (and finishes the tiny ones); without it they go through pass 2 one by one — the same results, and a
// leaner kernel for batches that hold none.  The host picks the variant from what the previous batch on the handle held.
template <int ITEMS, bool PAIRS>
__global__ __launch_bounds__(KMX_BLOCK, PAIRS ? KMX_LOOKUP_OCC : (ITEMS > 4 ? 4 : 5)) void k_lookup(const KmxIndexDev* __restrict__ ix,
                                                      const uint8_t* __restrict__ qranks,
                                                      const uint64_t* __restrict__ qoff, uint64_t nq,
                                                      QueryDesc d, unsigned long long* __restrict__ ctr,
                                                      uint64_t* __restrict__ block_hits, uint32_t flags)
{
    // block_hits[blockIdx.x] = sum of the block's hit counts: the first level of the scan that follows (k_scan_reduce's
    // job, for free here)
    __shared__ BlockCounters bc;
    __shared__ KmxElemDev elems_s[KMX_MAX_KS];      // the element descriptors: read at LDS latency, no vector-memory issue
    __shared__ uint32_t plan_s[KMX_PLAN_LDS];       // the planner's entries of the short lengths (second summand of a multi-k pair)
    static_assert(KMX_PLAN_LDS == KMX_BLOCK, "one planner entry per thread");
    if (threadIdx.x == 0) {
        bc.n_long = 0;
        bc.n_stitch = bc.n_stitch_tiny = bc.n_stitch_short = bc.n_resolved = bc.n_prefix = bc.n_prefix_big = bc.n_prefix_merge = bc.n_prefix_mid = bc.n_prefix_plain = bc.n_error = bc.n_none = bc.n_more = 0;
        bc.words = bc.pelems = bc.hits = 0;
        bc.max_runs = 0;
        unsigned int j = 0;
        while (j < 63 && ix->pw[j + 1] <= KMX_SUBK_FANOUT_LIMIT) ++j;
        bc.max_fan_exp = j;
        // hash16<true>: every partial Horner sum below 2^24 (sigma^(k-1) <= 2^24) and the hash below 2^32, for every element
        unsigned int small = ix->sigma < (1u << 24) ? 1u : 0u;
        for (uint32_t e = 0; e < ix->n_ks; ++e) {
            const uint64_t nk = ix->elems[e].n_keys;
            if (nk > (uint64_t(1) << 32) || nk / ix->sigma > (uint64_t(1) << 24)) small = 0;
        }
        bc.small_keys = small;
    }
    {
        const uint32_t n_words = ix->n_ks * uint32_t(sizeof(KmxElemDev) / 8);
        const uint64_t* __restrict__ srcw = reinterpret_cast<const uint64_t*>(ix->elems);
        uint64_t* dstw = reinterpret_cast<uint64_t*>(elems_s);
        for (uint32_t i = threadIdx.x; i < n_words; i += KMX_BLOCK) dstw[i] = srcw[i];
        plan_s[threadIdx.x] = threadIdx.x < ix->range ? ((const KMX_GLOBAL uint32_t*)ix->plan)[threadIdx.x] : 0u;
    }
    const uint8_t* __restrict__ qend = qranks + qoff[nq];
    __syncthreads();

    // a block serves ITEMS * 256 queries so that the (returning) global atomics of the
    // work-list bookkeeping are paid once per 1024 queries
    uint8_t kinds[ITEMS];
    unsigned int locs[ITEMS];
    unsigned long long locw[ITEMS];
    bool done[ITEMS];
    uint64_t my_hits = 0;
    const uint32_t sigma = ix->sigma, range = ix->range;
    const bool small_keys = __builtin_amdgcn_readfirstlane(int(bc.small_keys)) != 0;
    const uint8_t* __restrict__ dummy = reinterpret_cast<const uint8_t*>(ix);     // always >= 64 readable bytes

    unsigned int my_resolved = 0, my_err = 0, my_none = 0;     // (outcomes counted per thread, one LDS atomic each at the end)
    // One query's descriptor out, with the block-aggregated bookkeeping of the work lists (LDS counters now, one global atomic
    // per counter and block later).  Everything that is not a plain exact lookup ends here: STITCH queries are sorted into the
    // tiny / short / general classes of the validation kernels, PREFIX ones into the classes of the merge kernels.
    auto emit = [&](int it, uint64_t q, uint8_t kind, uint8_t status, uint64_t src, uint64_t aux, uint64_t key, uint64_t p1, uint32_t cnt,
                    uint32_t c0, bool resolved) {
        unsigned int loc = 0;
        unsigned long long loc_words = 0;
        if (kind == KMX_KIND_STITCH && !resolved) {
            const uint32_t pc = uint32_t(p1);
            if (min(c0, pc) <= KMX_VTINY && max(c0, pc) <= 2 * KMX_VTINY) {
                loc = atomicAdd(&bc.n_stitch_tiny, 1u) | 0x80000000u;    // tiny: one thread validates it, listed from the back
            } else if (!(p1 & (KMX_P1_MORE | KMX_P1_BIG)) && min(c0, pc) <= KMX_VSHORT && max(c0, pc) <= KMX_VSTAGE) {
                loc = atomicAdd(&bc.n_stitch_short, 1u) | 0x40000000u;   // short: a list of its own; the shorter bucket is the one walked
                if (pc < c0) p1 |= KMX_P1_SWAP;
            } else {
                if ((p1 & (KMX_P1_MORE | KMX_P1_BIG)) || stitch_is_wide(p1)) atomicAdd(&bc.n_more, 1u);
                loc = atomicAdd(&bc.n_stitch, 1u);
            }
            loc_words = atomicAdd(&bc.words, (unsigned long long)(uint64_t(c0) / 64 + 1));      // compressed_bitset.hpp:23
        }
        if (kind == KMX_KIND_PREFIX) {
            const uint32_t plen = cnt - uint32_t(__popcll(aux));
            if (c0 < 2 || plen < 2) {
                loc = 0xFFFFFFFFu;                                   // one list (a prefix level's, or the only key): in order as it lies, not listed
                atomicAdd(&bc.n_prefix_plain, 1u);
            } else if (KMX_PSORT_IS_SMALL(c0, plen)) {
                loc = atomicAdd(&bc.n_prefix, 1u);                   // small: listed from the front
                if (KMX_PSORT_IS_MERGE(c0, plen)) atomicAdd(&bc.n_prefix_merge, 1u);
            } else {
                loc = atomicAdd(&bc.n_prefix_big, 1u) | 0x80000000u; // mid / large: listed from the back
                if (plen <= KMX_PSORT_MID_CAP) atomicAdd(&bc.n_prefix_mid, 1u);
                if (plen > KMX_PSORT_BLOCK_CAP) {                    // large: chunks sorted in LDS, then merged in global memory
                    atomicAdd(&bc.pelems, (unsigned long long)plen);
                    atomicMax(&bc.max_runs, (plen + KMX_PSORT_BLOCK_CAP - 1) / KMX_PSORT_BLOCK_CAP);
                }
            }
        }
        my_resolved += resolved;
        my_err += status != KMX_Q_OK;
        my_none += status == KMX_Q_OK && kind == KMX_KIND_NONE;
        kinds[it] = !resolved ? kind : uint8_t(KMX_KIND_NONE);    // (only the work-list bookkeeping at the end reads this)
        locs[it] = loc;
        locw[it] = loc_words;
        my_hits += cnt;
        d.src[q] = (kind == KMX_KIND_STITCH && !resolved) ? (src | SRC_SLOW) : (kind == KMX_KIND_PREFIX ? (src | SRC_FLAGS) : src);
        d.cnt[q] = cnt;
        d.kind[q] = kind;
        d.status[q] = status;
        if (kind == KMX_KIND_STITCH) {
            if (!resolved) {                                      // (a resolved query is a plain copy from here on: nobody asks for its parts)
                d.c0[q] = c0;
                d.key[q] = key;
                d.p1[q] = p1;
            }
        } else if (kind == KMX_KIND_PREFIX) {
            d.c0[q] = c0;
            d.aux[q] = aux;
            d.key[q] = key;
        }
    };

    uint64_t qb[ITEMS];
    uint32_t qm[ITEMS];
    uint32_t praw[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const uint64_t q = (uint64_t(blockIdx.x) * ITEMS + it) * KMX_BLOCK + threadIdx.x;
        const uint64_t qq = q < nq ? q : nq - 1;
        qb[it] = qoff[qq];
        const uint64_t mlen = qoff[qq + 1] - qb[it];
        qm[it] = mlen > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(mlen);
        kinds[it] = KMX_KIND_NONE;
        locs[it] = 0;
        locw[it] = 0;
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it)
        praw[it] = plan_effective(((const KMX_GLOBAL uint32_t*)ix->plan)[min(qm[it], range - 1)]);

    // ---- pass 1: the plain exact lookup (m == k <= 16, :198-205) and cross-referenced queries of exactly TWO parts with
    // k <= 16 each — one k with k < m <= 2k (:207-298: the second part is the k-mer that ends the query), or a multi-k pair
    // (:515-555) — with the thread's queries INTERLEAVED: all offsets, then all plan entries, then all letter loads (both parts),
    // then all probes (both parts) — one memory round trip per phase instead of one per query, part and phase.
    {
        uint32_t eab[ITEMS];                    // element of the first part | element of the second << 8 | offset of the second << 16
        bool two[ITEMS];
        u32x4_a1 wa[ITEMS], wb[ITEMS];
        unsigned int n_none = 0, n_err = 0;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint64_t q = (uint64_t(blockIdx.x) * ITEMS + it) * KMX_BLOCK + threadIdx.x;
            const uint32_t m = qm[it], scheme = praw[it] & 0xFF, e = (praw[it] >> 8) & 0xFF;
            const uint32_t kb = elems_s[e].k;
            const bool live = q < nq && m > 0 && m < range;
            const bool one = live && scheme == KMX_SCHEME_SINGLE && kb == m && m <= 16 && qranks + qb[it] + 16 <= qend;
            bool t = false;
            uint32_t ea = e, off = 0;
            if (PAIRS && live && !one && m < KMX_PLAN_LDS) {
                if (scheme == KMX_SCHEME_SINGLE) {
                    const uint32_t rest = m < 2 * kb ? m - kb : 0u;                   // (m > kb checked below)
                    t = m > kb && m <= 2 * kb && kb <= 16 && (rest == 0 || kb - rest <= bc.max_fan_exp);   // (:119-122 via :234 stays with pass 2)
                    off = m - kb;
                } else if (scheme == KMX_SCHEME_MULTI && ((praw[it] >> 16) & KMX_PLAN_NPARTS_MASK) == 2 && m > kb) {
                    // _optimal_nk_sum[m] = _optimal_nk_sum[m - k_last] + [k_last] (:434-435): the first summand is plan[m - k_last]
                    const uint32_t p2 = plan_s[m - kb];
                    ea = (p2 >> 8) & 0xFF;
                    const uint32_t ka = elems_s[ea].k;
                    t = (p2 & 0xFF) != KMX_SCHEME_NONE && ka + kb == m && ka <= 16 && kb <= 16;
                    off = m - kb;
                }
                t = t && qranks + qb[it] + off + 16 <= qend;                          // (both 16-byte loads inside the letters)
            }
            done[it] = one || t;
            two[it] = t;
            eab[it] = ea | (e << 8) | (off << 16);
            wa[it] = *reinterpret_cast<const u32x4_a1*>(done[it] ? qranks + qb[it] : dummy);
            if (PAIRS) wb[it] = *reinterpret_cast<const u32x4_a1*>(t ? qranks + qb[it] + off : dummy);
        }
        uint64_t ha[ITEMS], hb[ITEMS];
        bool oka[ITEMS], okb[ITEMS];
        u32x4_a1 pa[ITEMS], pb[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const KmxElemDev* ela = &elems_s[eab[it] & 0xFF];
            const KmxElemDev* elb = &elems_s[(eab[it] >> 8) & 0xFF];
            okb[it] = true; hb[it] = 0;
            if (small_keys) {
                oka[it] = hash16<true>(wa[it], done[it] ? ela->k : 0u, sigma, ha[it]);
                if (PAIRS) okb[it] = hash16<true>(wb[it], two[it] ? elb->k : 0u, sigma, hb[it]);
            } else {
                oka[it] = hash16<false>(wa[it], done[it] ? ela->k : 0u, sigma, ha[it]);
                if (PAIRS) okb[it] = hash16<false>(wb[it], two[it] ? elb->k : 0u, sigma, hb[it]);
            }
            // first probe: dense -> the table entries of h; open -> the slot {key, off, cnt}; both as one 16-byte load
            pa[it] = *(const KMX_GLOBAL u32x4_a1*)((done[it] && oka[it]) ? probe_first_addr(ela, ha[it]) : reinterpret_cast<const char*>(dummy));
            if (PAIRS) pb[it] = *(const KMX_GLOBAL u32x4_a1*)((two[it] && okb[it]) ? probe_first_addr(elb, hb[it]) : reinterpret_cast<const char*>(dummy));
        }
        // The buckets are known.  An exact lookup leaves right away; a pair of the TINY class — the shorter bucket at most KMX_VTINY
        // positions, the longer at most 2 KMX_VTINY (large k: 24-letter reads as two 12-mers) — is finished HERE: both buckets are
        // read (32 + 64 bytes, straight-line — any element of the arena may be read 64 bytes wide, its allocation is padded), the
        // shorter list is looked up in the longer (membership is symmetric, whichever part it is), and when the surviving start
        // positions are one run of the first bucket the query leaves as a plain copy of that run — no descriptor round trip
        // through a validation kernel, no mask words (not with KEEP_MASKS: the words are the point there).  Two queries' bucket
        // loads are in flight at a time.
        struct Pend {
            bool two, tiny, first_is_short;
            uint8_t kind, status;
            uint64_t src, key, p1;
            uint32_t c0, off, pc;
            u32x4 s0, s1, l0, l1, l2, l3;
        };
        const KMX_GLOBAL uint32_t* ar = as_global(ix->arena);
        auto stage1 = [&](int it) -> Pend {
            Pend P{};
            P.two = false; P.tiny = false; P.first_is_short = true; P.kind = KMX_KIND_NONE; P.status = KMX_Q_OK; P.src = P.key = P.p1 = 0; P.c0 = P.off = P.pc = 0;
            const uint64_t q = (uint64_t(blockIdx.x) * ITEMS + it) * KMX_BLOCK + threadIdx.x;
            Run ra{0, 0}, rb{0, 0};
            if (done[it]) {
                const KmxElemDev* ela = &elems_s[eab[it] & 0xFF];
                if (oka[it]) ra = probe_finish(ela, ha[it], pa[it]);
                if (!PAIRS || !two[it]) {
                    const uint8_t status = oka[it] ? KMX_Q_OK : KMX_Q_BAD_RANK;
                    n_err += status != KMX_Q_OK;
                    n_none += status == KMX_Q_OK && !ra.cnt;
                    d.src[q] = ra.src;
                    d.cnt[q] = ra.cnt;
                    d.kind[q] = ra.cnt ? KMX_KIND_EXACT : KMX_KIND_NONE;
                    d.status[q] = status;
                    my_hits += ra.cnt;
                } else {
                    const KmxElemDev* elb = &elems_s[(eab[it] >> 8) & 0xFF];
                    const uint32_t off = eab[it] >> 16;
                    const bool multi = (praw[it] & 0xFF) == KMX_SCHEME_MULTI;
                    if (okb[it]) rb = probe_finish(elb, hb[it], pb[it]);
                    // the parts are looked at in the reference's order and a miss ends the walk (:216-227; :519-527 from the last
                    // summand): a letter outside the alphabet only counts in a part the walk reaches
                    const bool bad = multi ? (!okb[it] || (rb.cnt != 0 && !oka[it])) : (!oka[it] || (ra.cnt != 0 && !okb[it]));
                    P.two = true;
                    if (bad) {
                        P.status = KMX_Q_BAD_RANK;
                    } else if (ra.cnt != 0 && rb.cnt != 0) {
                        P.kind = KMX_KIND_STITCH; P.src = ra.src; P.c0 = ra.cnt; P.key = rb.src; P.off = off; P.pc = rb.cnt;
                        P.p1 = (uint64_t(off) << 32) | rb.cnt;
                        if (!(flags & KMX_SEARCH_KEEP_MASKS) && stitch_goes_big(ra.cnt, rb.cnt)) P.p1 |= KMX_P1_BIG;
                        P.tiny = !(flags & KMX_SEARCH_KEEP_MASKS) && min(ra.cnt, rb.cnt) <= KMX_VTINY && max(ra.cnt, rb.cnt) <= 2 * KMX_VTINY;
                        P.first_is_short = ra.cnt <= rb.cnt;
                    }
                }
            }
            static_assert(KMX_VTINY == 8, "two 16-byte loads for the shorter bucket, four for the longer");
            if (!PAIRS) return P;
            const uint64_t sa = !P.tiny ? 0 : P.first_is_short ? ra.src : rb.src, la = !P.tiny ? 0 : P.first_is_short ? rb.src : ra.src;
            P.s0 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + sa); P.s1 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + sa + 4);
            P.l0 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + la); P.l1 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + la + 4);
            P.l2 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + la + 8); P.l3 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + la + 12);
            return P;
        };
        auto stage2 = [&](int it, const Pend& P) {
            if (!PAIRS || !P.two) return;
            const uint64_t q = (uint64_t(blockIdx.x) * ITEMS + it) * KMX_BLOCK + threadIdx.x;
            uint64_t src = P.src;
            uint32_t cnt = 0;
            bool resolved = false;
            if (P.tiny) {
                // S: the shorter list, L: the longer.  An entry s of S meets the entry s + shift of L: shift = +off when S is the first
                // part's bucket, -off when it is the second's (s < off wraps past every position: n + k - 1 < 2^32, :169-170)
                const uint32_t sv[8] = {P.s0.x, P.s0.y, P.s0.z, P.s0.w, P.s1.x, P.s1.y, P.s1.z, P.s1.w};
                const uint32_t lv[16] = {P.l0.x, P.l0.y, P.l0.z, P.l0.w, P.l1.x, P.l1.y, P.l1.z, P.l1.w, P.l2.x, P.l2.y, P.l2.z, P.l2.w, P.l3.x, P.l3.y, P.l3.z, P.l3.w};
                const uint32_t n_s = P.first_is_short ? P.c0 : P.pc, n_l = P.first_is_short ? P.pc : P.c0;
                const uint32_t shift = P.first_is_short ? P.off : 0u - P.off;
                // (entries past a list's end never match: L's read 0xFFFFFFFF, S's look for 0xFFFFFFFE — neither is a position, a wrapped
                //  s - off is sent to 0xFFFFFFFE too; membership as the minimum of x ^ l over L: vector instructions only)
                uint32_t lm[16];
#pragma unroll
                for (uint32_t b = 0; b < 16; ++b) lm[b] = b < n_l ? lv[b] : 0xFFFFFFFFu;
                uint32_t met = 0;                                  // bit a: S[a] is met in L (binary_search :283, lower_bound :544-546)
                uint32_t last = 0;                                 // the last such entry, as a start position's partner in L
#pragma unroll
                for (uint32_t a = 0; a < 8; ++a) {
                    const bool usable = a < n_s && (P.first_is_short || sv[a] >= P.off);
                    const uint32_t x = usable ? sv[a] + shift : 0xFFFFFFFEu;
                    uint32_t diff = 0xFFFFFFFFu;
#pragma unroll
                    for (uint32_t b = 0; b < 16; b += 2) diff = min(diff, min(x ^ lm[b], x ^ lm[b + 1]));
                    const bool hit = diff == 0;
                    met |= uint32_t(hit) << a;
                    last = hit ? x : last;
                }
                const uint32_t n_met = uint32_t(__popc(met));
                if (P.first_is_short) {
                    const uint32_t lo = met ? uint32_t(__ffs(int(met))) - 1u : 0u;
                    if ((met >> lo) == (1u << n_met) - 1u) {       // one run of the first bucket (or nothing)
                        resolved = true; src = P.src + lo; cnt = n_met;
                    }
                } else if (n_met <= 1) {
                    // the walked list was the SECOND part's: the one survivor's place in the first bucket (= L) is its rank there
                    uint32_t rank = 0;
#pragma unroll
                    for (uint32_t b = 0; b < 16; ++b) rank += uint32_t(lm[b] < last);
                    resolved = true; src = P.src + (n_met ? rank : 0u); cnt = n_met;
                }
            }
            emit(it, q, P.kind, P.status, src, 0, P.key, P.p1, cnt, P.c0, resolved);
        };
        static_assert(ITEMS == 4 || ITEMS == 8, "the two-in-flight schedule below");
        {
            Pend p0 = stage1(0);
            Pend p1 = stage1(1);
#pragma unroll
            for (int it = 2; it < ITEMS; it += 2) {
                stage2(it - 2, p0);
                p0 = stage1(it);
                stage2(it - 1, p1);
                p1 = stage1(it + 1);
            }
            stage2(ITEMS - 2, p0);
            stage2(ITEMS - 1, p1);
        }
        if (n_err) atomicAdd(&bc.n_error, n_err);
        if (n_none) atomicAdd(&bc.n_none, n_none);
    }

    // ---- pass 2: everything else (other lengths, long queries, more parts), one query at a time
#pragma unroll 1
    for (int it = 0; it < ITEMS; ++it) {
        if (done[it]) continue;
        const uint64_t q = (uint64_t(blockIdx.x) * ITEMS + it) * KMX_BLOCK + threadIdx.x;
        if (q >= nq) continue;
        const uint64_t b = qoff[q];
        const uint64_t m = qoff[q + 1] - b;
        // A query of very many parts (a 5000-letter read on k = 10: 500 probes) walked by ONE lane is a chain of hundreds of
        // dependent round trips; such queries are counted, and — when the engine has launched k_lookup_long behind this kernel
        // (it does when the batch before held some) — only listed here: there a wave takes the query, a lane per part.
        if (m > 0 && m < range) {
            const KmxPlanEntry pe = plan_unpack(plan_effective(m < KMX_PLAN_LDS ? plan_s[m] : ((const KMX_GLOBAL uint32_t*)ix->plan)[m]));
            if (pe.scheme == KMX_SCHEME_SINGLE) {
                const uint32_t k = elems_s[pe.elem].k;
                if (m > uint64_t(k) * KMX_LONG_PARTS) {
                    const unsigned int slot = atomicAdd(&bc.n_long, 1u);
                    if (flags & KMX_SEARCH_INTERNAL_DEFER_LONG) {
                        kinds[it] = 0xFF;                        // (listed in the last phase, from the back of short_list)
                        locs[it] = slot;
                        continue;
                    }
                }
            }
        }
        LookupOut lo;
        lookup_query(ix, elems_s, qranks + b, m, qend, flags & ~KMX_SEARCH_INTERNAL_DEFER_LONG, lo);
        emit(it, q, lo.kind, lo.status, lo.src, lo.aux, lo.key, lo.p1, lo.cnt, lo.c0, lo.resolved);
    }

    if (my_resolved) atomicAdd(&bc.n_resolved, my_resolved);
    if (my_err) atomicAdd(&bc.n_error, my_err);
    if (my_none) atomicAdd(&bc.n_none, my_none);
    for (int off = 32; off > 0; off >>= 1) my_hits += __shfl_xor(my_hits, off);
    if (lane_id() == 0 && my_hits) atomicAdd(&bc.hits, (unsigned long long)my_hits);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (block_hits) block_hits[blockIdx.x] = bc.hits;
        if (bc.n_stitch | bc.n_stitch_tiny | bc.n_stitch_short) {
            if (bc.n_stitch) bc.base_stitch = (unsigned int)atomicAdd(&ctr[KMX_CTR_STITCH], (unsigned long long)bc.n_stitch);
            if (bc.n_stitch_tiny) bc.base_stitch_tiny = (unsigned int)atomicAdd(&ctr[KMX_CTR_STITCH_TINY], (unsigned long long)bc.n_stitch_tiny);
            if (bc.n_stitch_short) bc.base_stitch_short = (unsigned int)atomicAdd(&ctr[KMX_CTR_STITCH_SHORT], (unsigned long long)bc.n_stitch_short);
            bc.base_words = atomicAdd(&ctr[KMX_CTR_MASK_WORDS], bc.words);
            if (bc.n_more) atomicAdd(&ctr[KMX_CTR_STITCH_MORE], (unsigned long long)bc.n_more);
        }
        if (bc.n_prefix) bc.base_prefix = (unsigned int)atomicAdd(&ctr[KMX_CTR_PREFIX], (unsigned long long)bc.n_prefix);
        if (bc.n_prefix_merge) atomicAdd(&ctr[KMX_CTR_PREFIX_MERGE], (unsigned long long)bc.n_prefix_merge);
        if (bc.n_prefix_mid) atomicAdd(&ctr[KMX_CTR_PREFIX_MID], (unsigned long long)bc.n_prefix_mid);
        if (bc.n_prefix_plain) atomicAdd(&ctr[KMX_CTR_PREFIX_PLAIN], (unsigned long long)bc.n_prefix_plain);
        if (bc.n_prefix_big) {
            bc.base_prefix_big = (unsigned int)atomicAdd(&ctr[KMX_CTR_PREFIX_BIG], (unsigned long long)bc.n_prefix_big);
            if (bc.pelems) atomicAdd(&ctr[KMX_CTR_PREFIX_ELEMS], bc.pelems);
            if (bc.max_runs) atomicMax(&ctr[KMX_CTR_MAX_RUNS], (unsigned long long)bc.max_runs);
        }
        if (bc.n_long) bc.base_long = (unsigned int)atomicAdd(&ctr[KMX_CTR_LONG], (unsigned long long)bc.n_long);
        if (bc.n_resolved) atomicAdd(&ctr[KMX_CTR_STITCH_RESOLVED], (unsigned long long)bc.n_resolved);
        if (bc.n_error) atomicAdd(&ctr[KMX_CTR_ERROR], (unsigned long long)bc.n_error);
        if (bc.n_none) atomicAdd(&ctr[KMX_CTR_NONE], (unsigned long long)bc.n_none);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const uint64_t q = (uint64_t(blockIdx.x) * ITEMS + it) * KMX_BLOCK + threadIdx.x;
        if (kinds[it] == 0xFF) {
            d.short_list[nq - 1 - (bc.base_long + locs[it])] = uint32_t(q);      // deferred to k_lookup_long
        } else if (kinds[it] == KMX_KIND_STITCH) {
            d.aux[q] = bc.base_words + locw[it];              // first mask word of this query
            if (locs[it] & 0x80000000u) d.stitch_list[nq - 1 - (bc.base_stitch_tiny + (locs[it] & 0x3FFFFFFFu))] = uint32_t(q);
            else if (locs[it] & 0x40000000u) d.short_list[bc.base_stitch_short + (locs[it] & 0x3FFFFFFFu)] = uint32_t(q);
            else d.stitch_list[bc.base_stitch + locs[it]] = uint32_t(q);
        } else if (kinds[it] == KMX_KIND_PREFIX && locs[it] != 0xFFFFFFFFu) {
            if (locs[it] & 0x80000000u) d.prefix_list[nq - 1 - (bc.base_prefix_big + (locs[it] & 0x7FFFFFFFu))] = uint32_t(q);
            else d.prefix_list[bc.base_prefix + locs[it]] = uint32_t(q);
        }
    }
}

// ---------------------------------------------------------------------------
// k_lookup_long — single-k queries of more than KMX_LONG_PARTS parts that k_lookup listed instead of walking (:207-339 with
// hundreds of parts): one WAVE per query, a lane per part — rank-hash and probe of all parts in parallel (64 at a time), then
// the outcome of the reference's walk: it looks at the parts in order and stops at the first one that is missing (:221-224:
// an empty result) — a letter outside the alphabet only counts in a part it reaches — and throws for a rest whose prefix range
// is too wide (:119-122 via :234) only when every full part was found.  Lane 0 writes the descriptor and does the work-list
// bookkeeping of k_lookup's emit() with global atomics (these queries are few).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(KMX_BLOCK) void k_lookup_long(const KmxIndexDev* __restrict__ ix, const uint8_t* __restrict__ qranks,
                                                           const uint64_t* __restrict__ qoff, uint64_t nq, QueryDesc d,
                                                           unsigned long long* __restrict__ ctr, uint32_t flags)
{
    const uint32_t lane = lane_id();
    const uint64_t wave = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    const uint64_t n_waves = uint64_t(gridDim.x) * (KMX_BLOCK / KMX_WAVE);
    const uint64_t n_long = ctr[KMX_CTR_LONG];
    const uint8_t* __restrict__ qend = qranks + qoff[nq];
    const uint32_t sigma = ix->sigma;
    for (uint64_t i = wave; i < n_long; i += n_waves) {
        const uint32_t q = d.short_list[nq - 1 - i];
        const uint64_t b = qoff[q], m = qoff[q + 1] - b;
        const uint8_t* __restrict__ qr = qranks + b;
        const KmxPlanEntry pe = load_plan(ix, m);
        const KmxElemDev* __restrict__ el = &ix->elems[pe.elem];
        const uint32_t k = el->k;
        const uint32_t P = uint32_t(m / k), rest = uint32_t(m % k);
        const bool fan = rest && ix->pw[k - rest] > KMX_SUBK_FANOUT_LIMIT;
        const uint32_t n_parts = P + ((rest && !fan) ? 1u : 0u);                      // (the rest is only looked at when its lookup would not throw)
        uint32_t my_stop = 0xFFFFFFFFu;                                              // first part of this lane at which the walk would stop, x 2 + (1: missing, 0: bad letter)
        Run first{0, 0}, last{0, 0};
        for (uint32_t j = lane; j < n_parts; j += KMX_WAVE) {
            const uint64_t start = j < P ? uint64_t(j) * k : m - k;
            uint64_t h;
            const bool ok = rank_hash(qr + start, k, sigma, h, qend);
            Run r{0, 0};
            if (ok) r = probe(el, h);
            if ((!ok || r.cnt == 0) && my_stop == 0xFFFFFFFFu) my_stop = 2 * j + (ok ? 1u : 0u);
            if (j == 0) first = r;
            if (j == n_parts - 1) last = r;
        }
        uint32_t stop = my_stop;
        for (int off = 32; off > 0; off >>= 1) stop = min(stop, uint32_t(__shfl_xor(int(stop), off)));
        const int last_lane = int((n_parts - 1) % KMX_WAVE);
        const uint64_t last_src = __shfl(last.src, last_lane);
        const uint32_t last_cnt = uint32_t(__shfl(int(last.cnt), last_lane));
        if (lane != 0) continue;
        uint8_t kind = KMX_KIND_NONE, status = KMX_Q_OK;
        if (stop != 0xFFFFFFFFu) {
            if (!(stop & 1u)) status = KMX_Q_BAD_RANK;
        } else if (fan) {
            status = KMX_Q_SUBK_FANOUT;
        } else {
            kind = KMX_KIND_STITCH;
        }
        d.cnt[q] = 0;
        d.kind[q] = kind;
        d.status[q] = status;
        if (kind != KMX_KIND_STITCH) {
            d.src[q] = 0;
            atomicAdd(&ctr[status != KMX_Q_OK ? KMX_CTR_ERROR : KMX_CTR_NONE], 1ull);
            continue;
        }
        const uint32_t c0 = first.cnt, delta = uint32_t(rest ? m - k : uint64_t(P - 1) * k);
        uint64_t p1 = (uint64_t(delta) << 32) | last_cnt | KMX_P1_MORE;                // (more than KMX_LONG_PARTS parts: always further ones)
        if (!(flags & KMX_SEARCH_KEEP_MASKS) && stitch_goes_big(c0, last_cnt)) p1 |= KMX_P1_BIG;
        d.src[q] = first.src | SRC_SLOW;
        d.c0[q] = c0;
        d.key[q] = last_src;
        d.p1[q] = p1;
        d.aux[q] = atomicAdd(&ctr[KMX_CTR_MASK_WORDS], (unsigned long long)(uint64_t(c0) / 64 + 1));     // compressed_bitset.hpp:23
        if (min(c0, last_cnt) <= KMX_VTINY && max(c0, last_cnt) <= 2 * KMX_VTINY) {
            d.stitch_list[nq - 1 - atomicAdd(&ctr[KMX_CTR_STITCH_TINY], 1ull)] = q;
        } else {
            atomicAdd(&ctr[KMX_CTR_STITCH_MORE], 1ull);
            d.stitch_list[atomicAdd(&ctr[KMX_CTR_STITCH], 1ull)] = q;
        }
    }
}

void launch_lookup_long(hipStream_t s, const KmxIndexDev* ix, const uint8_t* qranks, const uint64_t* qoff, uint64_t nq, const QueryDesc& d,
                        unsigned long long* ctr, uint32_t flags)
{
    // (the number of listed queries is on the device: a fixed grid, every wave strides over the list)
    hipLaunchKernelGGL(k_lookup_long, dim3(256 * 4), dim3(KMX_BLOCK), 0, s, ix, qranks, qoff, nq, d, ctr, flags);
}

// ---------------------------------------------------------------------------
// k_validate — STITCH queries.
//
// Candidates are the positions of the first part's bucket (kmer_index.hpp:272,
// :532).  Candidate p survives when, for every further part j that starts at
// query offset s_j, p + s_j is a member of part j's bucket (:279-291, :541-551).
// A rest that is shorter than k (:230-256) is checked through the k-mer that ends
// the query (offset m - k) instead of through the sigma^(k-rest) prefix buckets:
// both decide "the text continues with the rest of the query at p + P*k".
//
// KMX_VGROUPS queries per wave, one per KMX_VGROUP-lane group.  Every STITCH query names one further part
// (QueryDesc::key / p1): its bucket is staged in LDS and filters all candidates at LDS latency.  A query with
// more parts than that checks them for the survivors of the filter only — they are few: a candidate passes
// by chance with probability bucket/n — one part per lane of the group.  64 candidates = one
// compressed_bitset word, assembled from 16-bit slices of the wave's ballots.
// ---------------------------------------------------------------------------
// Does candidate position p continue with every further part of query q?  Called by a whole KMX_VGROUP-lane
// group (gl = lane in group); lane gl takes parts gl, gl + 16, ...  Part e of a single-k query: the k-part at
// (e + 1) * k (:279-291), last the k-mer that ends the query when there is a rest; of a multi-k query: the
// e-th summand of _optimal_nk_sum[m] counted from the end (:532-555).  Returns this lane's verdict.
__device__ __forceinline__ bool stitch_parts_hold(const KmxIndexDev* __restrict__ ix, const uint32_t* __restrict__ arena,
                                                  const uint8_t* __restrict__ qranks, const uint64_t* __restrict__ qoff,
                                                  uint32_t q, uint32_t p, uint32_t gl, uint32_t stride = KMX_VGROUP,
                                                  uint32_t skip_start = 0xFFFFFFFFu)
{
    const uint64_t b = qoff[q];
    const uint64_t m = qoff[q + 1] - b;
    const uint8_t* __restrict__ qr = qranks + b;
    const uint8_t* __restrict__ qend = qranks + qoff[q + 1];
    const uint32_t sigma = ix->sigma;
    const KmxPlanEntry pe = load_plan(ix, m);
    bool good = true;
    if (pe.scheme == KMX_SCHEME_SINGLE) {
        const KmxElemDev* __restrict__ el = &ix->elems[pe.elem];
        const uint32_t k = el->k;
        const uint32_t P = uint32_t(m / k);
        const uint32_t n_extra = P - 1 + ((m % k) ? 1 : 0);
        for (uint32_t e = gl; e < n_extra && good; e += stride) {
            const uint64_t start = (e < P - 1) ? uint64_t(e + 1) * k : (m - k);
            if (uint32_t(start) == skip_start) continue;               // (the caller has this part's verdict already)
            uint64_t h;
            rank_hash(qr + start, k, sigma, h, qend);
            const Run r = probe(el, h);
            const uint32_t x = p + uint32_t(start);
            const uint64_t lb = lower_bound_dev<uint32_t>(arena + r.src, r.cnt, x);
            good = lb < r.cnt && arena[r.src + lb] == x;              // binary_search :283
        }
    } else {
        const uint32_t n_extra = pe.nparts - 1u;
        uint64_t mm = m;                                               // the walk resumes where this lane stopped
        uint32_t walked = 0;
        for (uint32_t e = gl; e < n_extra && good; e += stride) {
            const KmxElemDev* __restrict__ el = nullptr;
            uint32_t k = 0;
            for (; walked <= e; ++walked) {
                const KmxPlanEntry en = load_plan_chain(ix, mm);
                el = &ix->elems[en.elem];
                k = el->k;
                mm -= k;                                               // this summand covers [mm, mm + k)
            }
            if (uint32_t(mm) == skip_start) continue;
            uint64_t h;
            rank_hash(qr + mm, k, sigma, h, qend);
            const Run r = probe(el, h);                                // search_k, :520
            const uint32_t x = p + uint32_t(mm);
            const uint64_t lb = lower_bound_dev<uint32_t>(arena + r.src, r.cnt, x);
            good = lb < r.cnt && arena[r.src + lb] == x;              // lower_bound :544-546
        }
    }
    return good;
}

// Membership of NR values per lane in the sorted bucket staged at `arr` (padded with 0xFFFFFFFF to the power of two P):
// lower_bound (:283 binary_search, :544-546) as log2(P) branch-free halving steps, the NR searches in lockstep so that
// their LDS reads overlap.  Bit r of the result: x[r] is in the bucket.
template <int NR>
__device__ __forceinline__ uint32_t staged_members(const uint32_t* __restrict__ arr, uint32_t P, const uint32_t (&x)[KMX_VCH])
{
    uint32_t pos[NR], tv[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) pos[r] = 0;
    for (uint32_t st = P >> 1; st; st >>= 1) {
        const uint32_t* __restrict__ probe_at = arr + (st - 1);
#pragma unroll
        for (int r = 0; r < NR; ++r) tv[r] = probe_at[pos[r]];
#pragma unroll
        for (int r = 0; r < NR; ++r) pos[r] += tv[r] < x[r] ? st : 0u;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) tv[r] = arr[pos[r]];
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) m |= uint32_t(tv[r] == x[r]) << r;
    return m;
}

// INLINE_MORE: the survivors' further parts are checked right here (more registers, and the groups of a wave wait
// for each other's survivors); otherwise k_validate_more does it afterwards from the survivor lists.
template <bool INLINE_MORE>
__global__ __launch_bounds__(KMX_BLOCK) void k_validate(const KmxIndexDev* __restrict__ ix,
                                                        const uint32_t* __restrict__ arena,
                                                        const uint8_t* __restrict__ qranks,
                                                        const uint64_t* __restrict__ qoff, QueryDesc d,
                                                        uint64_t n_stitch, uint64_t* __restrict__ mask_words)
{
    __shared__ __attribute__((aligned(16))) uint32_t stage[KMX_BLOCK / KMX_WAVE][KMX_VGROUPS][KMX_VSTAGE];
    const uint32_t lane = lane_id();
    const uint32_t wv = threadIdx.x / KMX_WAVE;
    const uint32_t g = lane / KMX_VGROUP, gl = lane % KMX_VGROUP;
    static_assert(KMX_VGROUP == 16 && KMX_VGROUPS == 4, "the mask word assembly below is written for four 16-lane groups");
    const uint64_t wave = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    const uint64_t n_waves = uint64_t(gridDim.x) * (KMX_BLOCK / KMX_WAVE);

    for (uint64_t i0 = wave * KMX_VGROUPS; i0 < n_stitch; i0 += n_waves * KMX_VGROUPS) {
        const uint64_t i = i0 + g;
        const bool have = i < n_stitch;
        // straight-line loads: a group past the end of the list reads the list's last query and is masked by `have`
        const uint32_t q = d.stitch_list[min(i, n_stitch - 1)];
        // one round of independent loads per group
        const uint32_t c0 = d.c0[q];
        const uint64_t src_raw = d.src[q];
        const uint64_t p1 = d.p1[q];
        const uint64_t p1src = d.key[q];
        const uint64_t wbase = d.aux[q];
        const uint64_t src = src_raw & ~SRC_FLAGS;
        uint64_t* __restrict__ words = mask_words + wbase;
        const uint64_t sbase = wbase * 64;
        const bool fast = have && (INLINE_MORE || (!(p1 & KMX_P1_BIG) && !stitch_is_wide(p1)));  // (big: validate_big_wave; wide: k_validate_wide)
        const bool more = INLINE_MORE && (p1 & KMX_P1_MORE) != 0;
        const uint32_t pcnt = uint32_t(p1), delta = uint32_t(p1 >> 32) & KMX_P1_DELTA_MASK;
        const bool staged = fast && pcnt <= KMX_VSTAGE;
        // The staged bucket is padded with 0xFFFFFFFF (never a position) to the wave's largest power of
        // two, so that the search below is a fixed number of branch-free halving steps for all four groups.
        uint32_t P = (staged && pcnt > 1) ? (1u << (32 - __clz(int(pcnt - 1)))) : 1u;
        uint32_t max_it = fast ? (c0 + KMX_VGROUP - 1) / KMX_VGROUP : 0u;
        const uint32_t n_it = max_it;
#pragma unroll
        for (int e = 1; e < KMX_VGROUPS; ++e) {
            P = max(P, uint32_t(__shfl_xor(int(P), e * KMX_VGROUP)));
            max_it = max(max_it, uint32_t(__shfl_xor(int(max_it), e * KMX_VGROUP)));
        }
        P = uint32_t(__builtin_amdgcn_readfirstlane(int(P)));
        max_it = uint32_t(__builtin_amdgcn_readfirstlane(int(max_it)));
        uint32_t* __restrict__ arr = stage[wv][g];
        {
            // four consecutive entries per lane and step: one 16-byte load (any element of the arena may be read 16 bytes
            // wide: its allocation is padded), one 16-byte LDS store; a lane past the bucket re-reads its last entry.
            // Whole 64-entry steps are written (a group's array has room for KMX_VSTAGE = 256 >= P rounded up to 64).
            const uint32_t nb = staged ? pcnt : 0u, last = nb ? nb - 1u : 0u;
            const uint32_t* __restrict__ fil = arena + p1src;
            for (uint32_t t0 = 0; t0 < P; t0 += 4 * KMX_VGROUP) {
                const uint32_t t = t0 + 4u * gl;
                u32x4 v = *reinterpret_cast<const u32x4_a4*>(fil + min(t, last));
                v.x = t + 0 < nb ? v.x : 0xFFFFFFFFu;
                v.y = t + 1 < nb ? v.y : 0xFFFFFFFFu;
                v.z = t + 2 < nb ? v.z : 0xFFFFFFFFu;
                v.w = t + 3 < nb ? v.w : 0xFFFFFFFFu;
                *reinterpret_cast<u32x4*>(arr + t) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        uint32_t valid = 0;
        const uint32_t lt_mask = (1u << gl) - 1u;
        const uint32_t* __restrict__ cand = arena + src;
        const uint32_t c_last = (fast && c0) ? c0 - 1u : 0u;
        const bool unstaged = fast && !staged;
        // KMX_VCH rounds of 16 candidates at a time: their loads go out together and their searches advance
        // in lockstep, so one global and log2(P) + 1 LDS round trips are exposed per chunk instead of per round
        for (uint32_t it0 = 0; it0 < max_it; it0 += KMX_VCH) {
            const uint32_t nr = min(uint32_t(KMX_VCH), max_it - it0);  // wave-uniform
            const uint32_t ci0 = it0 * KMX_VGROUP + gl;
            uint32_t x[KMX_VCH];
            // one pointer per lane and chunk, clamped to the bucket's last entry; the rounds are immediate offsets from it
            // (a dead slot reads at most 7 x 16 entries past the bucket: inside the arena's KMX_ARENA_PAD)
            static_assert((KMX_VCH - 1) * KMX_VGROUP * 4 + 16 <= KMX_ARENA_PAD, "the dead slots of a chunk must stay inside the arena's padding");
            const uint32_t* __restrict__ cp = cand + min(ci0, c_last);
#pragma unroll
            for (int r = 0; r < KMX_VCH; ++r) x[r] = cp[r * KMX_VGROUP];
#pragma unroll
            for (int r = 0; r < KMX_VCH; ++r) x[r] += delta;
            // bit r: this lane has a candidate in round r
            const uint32_t n_live = (fast && ci0 < c0) ? min(uint32_t(KMX_VCH), (c0 - ci0 + KMX_VGROUP - 1) / KMX_VGROUP) : 0u;
            const uint32_t livem = (1u << n_live) - 1u;
            uint32_t okm;                                               // bit r: this lane's candidate of round r holds so far
            switch ((nr + 1) >> 1) {                                    // (straight-line code for 2, 4, 6 or 8 rounds)
                case 1: okm = staged_members<2>(arr, P, x); break;
                case 2: okm = staged_members<4>(arr, P, x); break;
                case 3: okm = staged_members<6>(arr, P, x); break;
                default: okm = staged_members<8>(arr, P, x); break;
            }
            okm &= livem;
            if (__any(unstaged)) {                                      // a filter bucket too long for LDS: searched where it lies
                if (unstaged) {
                    okm = 0;
                    for (uint32_t r = 0; r < n_live; ++r) {
                        const uint64_t lb = lower_bound_dev<uint32_t>(arena + p1src, pcnt, x[r]);
                        okm |= uint32_t(lb < pcnt && arena[p1src + lb] == x[r]) << r;
                    }
                }
            }
            if (INLINE_MORE && __any(more && okm != 0)) {
                // queries with further parts: the survivors of the filter, one per group at a time
#pragma unroll 1
                for (uint32_t r = 0; r < nr; ++r) {
                    uint32_t pend = uint32_t(__ballot(more && ((okm >> r) & 1u)) >> (KMX_VGROUP * g)) & 0xFFFFu;   // group-uniform
                    uint32_t dropped = 0;
                    while (__any(pend != 0)) {
                        if (pend) {
                            const uint32_t bsel = uint32_t(__ffs(int(pend))) - 1u;
                            const uint32_t pc = arena[src + uint64_t(it0 + r) * KMX_VGROUP + bsel];
                            const bool good = stitch_parts_hold(ix, arena, qranks, qoff, q, pc, gl);
                            const uint32_t verdicts = uint32_t(__ballot(good) >> (KMX_VGROUP * g)) & 0xFFFFu;
                            if (verdicts != 0xFFFFu) dropped |= 1u << bsel;
                            pend &= pend - 1;
                        }
                    }
                    if ((dropped >> gl) & 1u) okm &= ~(1u << r);
                }
            }
            // 64 candidates = one bitset word = four rounds of 16-bit ballot slices; a chunk of KMX_VCH = 8 rounds is two
            // words, and it0 is a multiple of 8, so which half-word a round fills is known at compile time
            static_assert(KMX_VCH == 8, "the two-words-per-chunk assembly below");
            uint32_t wq[4] = {0u, 0u, 0u, 0u};                          // {word A low, A high, word B low, B high}
            uint32_t* __restrict__ sh_out = d.stitch_hits ? d.stitch_hits + sbase : nullptr;
#pragma unroll
            for (int r = 0; r < KMX_VCH; ++r) {
                if (uint32_t(r) >= nr) break;
                const bool ok = (okm & (1u << r)) != 0;
                const uint32_t s16 = uint32_t(__ballot(ok) >> (KMX_VGROUP * g)) & 0xFFFFu;
                // survivors, already compacted and ascending: what k_fill copies out for this query
                if (ok && sh_out) sh_out[valid + uint32_t(__popc(s16 & lt_mask))] = x[r] - delta;
                valid += uint32_t(__popc(s16));
                wq[r >> 1] |= s16 << ((r & 1) * 16);
            }
            if (gl == 0) {                                              // bit i = word i>>6, bit i&63
                if (it0 < n_it) words[it0 >> 2] = (uint64_t(wq[1]) << 32) | wq[0];
                if (it0 + 4 < n_it) words[(it0 >> 2) + 1] = (uint64_t(wq[3]) << 32) | wq[2];
            }
        }
        if (fast && gl == 0) {
            if ((c0 & 63) == 0) words[c0 / 64] = 0;                     // n_bits/64 + 1 words (compressed_bitset.hpp:23)
            d.cnt[q] = valid;
        }
        __builtin_amdgcn_wave_barrier();                                // stage[] is reused by the next round

    }
}

// k_validate_short — STITCH queries of exactly TWO parts whose SHORTER bucket has at most KMX_VSHORT entries and whose longer
// one fits a group's stage (a 10-mer + 12-mer pair on a 1e8-letter text: 95 positions against 6).  Membership is symmetric:
// start position p survives when p is in the first part's bucket and p + offset in the second's (:279-291, :541-551) — so the
// SHORTER list is the one walked (one entry per lane of the 16-lane group, ONE lockstep search) and the longer one staged,
// whichever of the two is the first part (KMX_P1_SWAP: the walked entries are the second part's, the mask bit of a survivor
// is its index in the first part's bucket, which the search has just found).  Four queries per wave.
__global__ __launch_bounds__(KMX_BLOCK) void k_validate_short(const uint32_t* __restrict__ arena, QueryDesc d, uint64_t n_short,
                                                              uint64_t* __restrict__ mask_words)
{
    constexpr uint32_t MW = KMX_VSTAGE / 64 + 1;                        // mask words of a staged first bucket
    __shared__ __attribute__((aligned(16))) uint32_t stage[KMX_BLOCK / KMX_WAVE][KMX_VGROUPS][KMX_VSTAGE];
    __shared__ unsigned long long swords[KMX_BLOCK / KMX_WAVE][KMX_VGROUPS][MW + 3];
    const uint32_t lane = lane_id();
    const uint32_t wv = threadIdx.x / KMX_WAVE;
    const uint32_t g = lane / KMX_VGROUP, gl = lane % KMX_VGROUP;
    const uint64_t wave = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    const uint64_t n_waves = uint64_t(gridDim.x) * (KMX_BLOCK / KMX_WAVE);
    const uint32_t lt_mask = (1u << gl) - 1u;
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // (software-pipelined like k_validate: the next round's descriptor and the list entry behind it load during this round)
    const uint64_t stride = n_waves * KMX_VGROUPS;
    auto list_at = [&](uint64_t i) -> uint32_t { return d.short_list[min(i, n_short - 1)]; };
    uint32_t q_next = list_at(wave * KMX_VGROUPS + g);
    uint32_t q_after = list_at(wave * KMX_VGROUPS + stride + g);
    uint32_t n_c0 = d.c0[q_next];
    uint64_t n_src = d.src[q_next], n_p1 = d.p1[q_next], n_key = d.key[q_next], n_aux = d.aux[q_next];
    for (uint64_t i0 = wave * KMX_VGROUPS; i0 < n_short; i0 += stride) {
        const uint64_t i = i0 + g;
        const bool have = i < n_short;
        const uint32_t q = q_next;
        const uint32_t c0 = n_c0;
        const uint64_t src = n_src & ~SRC_FLAGS;
        const uint64_t p1 = n_p1;
        const uint64_t p1src = n_key;
        const uint64_t wbase = n_aux;
        q_next = q_after;
        n_c0 = d.c0[q_next]; n_src = d.src[q_next]; n_p1 = d.p1[q_next]; n_key = d.key[q_next]; n_aux = d.aux[q_next];
        q_after = list_at(i + 2 * stride);
        const uint32_t pcnt = uint32_t(p1), delta = uint32_t(p1 >> 32) & KMX_P1_DELTA_MASK;
        const bool swap = (p1 & KMX_P1_SWAP) != 0;
        // walked: n_w <= 16 entries at w_src; staged: n_s <= KMX_VSTAGE entries at s_src
        const uint64_t w_src = swap ? p1src : src, s_src = swap ? src : p1src;
        const uint32_t n_w = have ? (swap ? pcnt : c0) : 0u, n_s = have ? (swap ? c0 : pcnt) : 0u;
        uint32_t P = n_s > 1 ? (1u << (32 - __clz(int(n_s - 1)))) : 1u;
        P = max(P, uint32_t(__shfl_xor(int(P), KMX_VGROUP)));
        P = max(P, uint32_t(__shfl_xor(int(P), 2 * KMX_VGROUP)));
        P = uint32_t(__builtin_amdgcn_readfirstlane(int(P)));
        uint32_t* __restrict__ arr = stage[wv][g];
        // (the walked entry of this lane is asked for before the other bucket is staged: both travel together)
        const uint32_t e = arena[w_src + min(gl, n_w ? n_w - 1u : 0u)];
        {
            const uint32_t last = n_s ? n_s - 1u : 0u;
            const uint32_t* __restrict__ fil = arena + s_src;
            for (uint32_t t0 = 0; t0 < P; t0 += 4 * KMX_VGROUP) {
                const uint32_t t = t0 + 4u * gl;
                u32x4 v = *reinterpret_cast<const u32x4_a4*>(fil + min(t, last));
                v.x = t + 0 < n_s ? v.x : 0xFFFFFFFFu;
                v.y = t + 1 < n_s ? v.y : 0xFFFFFFFFu;
                v.z = t + 2 < n_s ? v.z : 0xFFFFFFFFu;
                v.w = t + 3 < n_s ? v.w : 0xFFFFFFFFu;
                *reinterpret_cast<u32x4*>(arr + t) = v;
            }
        }
        // the walked entry of this lane as a start position p, and what must be found in the staged bucket
        const bool live = gl < n_w && (!swap || e >= delta);            // (a start before the text is none)
        const uint32_t p = swap ? e - delta : e;
        const uint32_t x = swap ? p : e + delta;
        unsigned long long* sw = swords[wv][g];
        if (gl < MW) sw[gl] = 0;
        wsync();
        uint32_t pos = 0;
        for (uint32_t st = P >> 1; st; st >>= 1) pos += arr[pos + st - 1] < x ? st : 0u;      // lower_bound :544-546 / binary_search :283
        const bool ok = live && arr[pos] == x;
        const uint32_t s16 = uint32_t(__ballot(ok) >> (KMX_VGROUP * g)) & 0xFFFFu;
        // survivors ascend with the walked list; bit i of the mask = candidate i of the FIRST part's bucket
        if (ok && d.stitch_hits) d.stitch_hits[wbase * 64 + uint32_t(__popc(s16 & lt_mask))] = p;
        if (swap) {
            if (ok) atomicOr(&sw[pos >> 6], 1ull << (pos & 63));
            wsync();
            if (have && gl <= c0 / 64) mask_words[wbase + gl] = sw[gl];            // n_bits/64 + 1 words (compressed_bitset.hpp:23)
        } else if (have && gl == 0) {
            mask_words[wbase] = s16;                                               // c0 <= 16 candidates: one word
        }
        if (have && gl == 0) d.cnt[q] = uint32_t(__popc(s16));
        wsync();                                                        // stage[] / swords[] are the next round's
    }
}

#define KMX_VWIDE_SCAN 8
// k_validate_wide — STITCH queries whose filter bucket has 257 ... KMX_VWIDE entries: more than a 16-lane group of k_validate
// stages, few enough for the stage of a whole wave.  Candidates and filter bucket are both ASCENDING lists of about the same
// length here (two long buckets of a small k: 1526 positions per 8-mer at 1e8 letters), so the filter (:283 binary_search,
// :544-546 lower_bound per candidate) is done as a linear intersection: a lane takes E CONSECUTIVE candidates, finds the
// first one's place in the staged bucket by one binary search and WALKS from there — the next candidate's lower bound lies a
// few entries further on (a window of W entries, two or three halving steps, repeated in the rare case the window was too
// short) — instead of log2(bucket) probe steps for every candidate.  A lane's E verdicts are E consecutive bits of the
// compressed_bitset (compressed_bitset.hpp:13-14): E / 8 bytes of a byte array in LDS that is read back as 64-bit words.
// Lane = list entry while looking for such queries, then the wave takes them one by one.  Runs behind k_validate<false> and in
// front of the kernels that check further parts (they read the survivors this one leaves).
static_assert(KMX_VWIDE_MIN == KMX_VSTAGE, "k_validate stages filter buckets up to KMX_VSTAGE entries itself");
#define KMX_VWIDE_TILE 2048                  // candidates per pass of the wave (64 lanes x up to 32)
#define KMX_VWIDE_PAD 160                    // the bucket's stage ends in >= 64 entries of 0xFFFFFFFF (the walk's window); one more slot per 32 entries
template <int E>
__device__ __forceinline__ uint32_t wide_tile(const uint32_t* __restrict__ cand, uint32_t n_c, const uint32_t* __restrict__ fil, uint32_t pcnt,
                                              uint32_t P, uint32_t delta, uint64_t* __restrict__ words,
                                              uint32_t* __restrict__ sh_out, uint32_t valid, uint32_t* __restrict__ flat)
{
    static_assert((E == 8 || E == 16 || E == 32), "a lane's verdicts are whole bytes of the mask; E divides the wave");
    const uint32_t lane = lane_id();
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // 1. the tile's candidates (+ delta) through LDS: coalesced in, E consecutive ones per lane out (chunks E + 1 words apart:
    //    no bank conflicts); a slot past the last candidate holds 0, which never moves the walk
    //    (one pointer per lane, the rounds at immediate offsets: a slot past the bucket reads on into the arena — its allocation
    //    is padded by more than a tile — and is zeroed)
    static_assert(KMX_VWIDE_TILE * 4 <= KMX_ARENA_PAD, "a tile of candidates read past the last bucket must stay inside the arena's padding");
    uint32_t a[E];
    {
        const uint32_t* __restrict__ cp = cand + lane;
#pragma unroll
        for (int j = 0; j < E; ++j) a[j] = cp[j * KMX_WAVE];
        // candidate t = 64 j + lane belongs at t + t / E = (lane + lane / E) + j (64 + 64 / E): one address, immediate offsets
        uint32_t* __restrict__ wp = flat + (lane + lane / E);
#pragma unroll
        for (int j = 0; j < E; ++j) wp[j * (KMX_WAVE + KMX_WAVE / E)] = uint32_t(j) * KMX_WAVE + lane < n_c ? a[j] + delta : 0u;
    }
    wsync();
#pragma unroll
    for (int j = 0; j < E; ++j) a[j] = flat[lane * (E + 1) + j];
    wsync();
    // 2. the filter bucket, padded with 0xFFFFFFFF (never a position) to P + 64 entries, in a SKEWED layout: entry i at
    //    i + i / 32, and the free slot behind every 32 entries repeats the entry after it.  The array stays ascending (a
    //    lower bound and a membership test do not mind a repeat), and the lanes — whose places in the bucket lie about 32
    //    entries apart, one chunk of candidates each — meet different LDS banks instead of two.
    {
        const uint32_t last = pcnt - 1u;
        for (uint32_t t0 = 0; t0 < P + 64; t0 += 4 * KMX_WAVE) {
            const uint32_t t = t0 + 4u * lane;
            u32x4 v = *reinterpret_cast<const u32x4_a4*>(fil + min(t, last));
            v.x = t + 0 < pcnt ? v.x : 0xFFFFFFFFu;
            v.y = t + 1 < pcnt ? v.y : 0xFFFFFFFFu;
            v.z = t + 2 < pcnt ? v.z : 0xFFFFFFFFu;
            v.w = t + 3 < pcnt ? v.w : 0xFFFFFFFFu;
            if (t < P + 64) {
                uint32_t* __restrict__ o = flat + (t + (t >> 5));          // (four entries never straddle a group of 32)
                o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
                if ((t & 31u) == 0 && t) o[-1] = v.x;                     // the repeat in front of a group
            }
        }
    }
    wsync();
    // 3. where the first candidate of a chain belongs (log2 P halving steps over the entries), then the walk over the skewed
    //    array: a window of eight slots, three halving steps at immediate offsets, once more when a chain's window was too
    //    short — as C = 4 independent chains per lane (quarters of its E candidates), advanced in lockstep: a chain is one
    //    dependent LDS read after the other, four of them keep the reads of a lane overlapping
    const KMX_LDS uint32_t* B = (const KMX_LDS uint32_t*)flat;
    constexpr int C = 4, LEN = E / C;
    uint32_t pb[C], tv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) pb[c] = 0;
    for (uint32_t st = P >> 1; st; st >>= 1) {
#pragma unroll
        for (int c = 0; c < C; ++c) { const uint32_t i = pb[c] + st - 1; tv[c] = B[i + (i >> 5)]; }
#pragma unroll
        for (int c = 0; c < C; ++c) pb[c] += tv[c] < a[c * LEN] ? st : 0u;
    }
    const KMX_LDS uint32_t* pp[C];                                        // the chains' places in the skewed array
#pragma unroll
    for (int c = 0; c < C; ++c) pp[c] = B + (pb[c] + (pb[c] >> 5));
    uint32_t res = 0;
#pragma unroll
    for (int j = 0; j < LEN; ++j) {
        for (;;) {
#pragma unroll
            for (int st = 4; st; st >>= 1) {
#pragma unroll
                for (int c = 0; c < C; ++c) tv[c] = pp[c][st - 1];
#pragma unroll
                for (int c = 0; c < C; ++c) pp[c] += tv[c] < a[c * LEN + j] ? st : 0;
            }
#pragma unroll
            for (int c = 0; c < C; ++c) tv[c] = *pp[c];
            bool more = false;
#pragma unroll
            for (int c = 0; c < C; ++c) more |= tv[c] < a[c * LEN + j];
            if (!__any(more)) break;                                   // (the window was too short for some chain: once more)
        }
#pragma unroll
        for (int c = 0; c < C; ++c) res |= uint32_t(tv[c] == a[c * LEN + j]) << (c * LEN + j);
    }
    const uint32_t first = lane * E;
    const uint32_t live = first < n_c ? min(uint32_t(E), n_c - first) : 0u;
    res &= live >= 32 ? 0xFFFFFFFFu : (1u << live) - 1u;
    wsync();                                                           // every read of the bucket is done: the stage takes the verdicts
    // 4. bit i of the mask = candidate i: the lanes' verdicts as bytes, the words read back
    {
        KMX_LDS uint8_t* rb = (KMX_LDS uint8_t*)flat;
#pragma unroll
        for (int by = 0; by < E / 8; ++by) rb[lane * (E / 8) + by] = uint8_t(res >> (8 * by));
        wsync();
        const uint32_t n_words = (n_c + 63) / 64;
        if (lane < n_words) words[lane] = ((const KMX_LDS uint64_t*)flat)[lane];
    }
    // 5. the survivors, compacted and ascending: what k_fill copies out for this query
    const uint32_t mine = uint32_t(__popc(res));
    uint32_t incl = mine;
#pragma unroll
    for (uint32_t o = 1; o < KMX_WAVE; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const uint32_t total = uint32_t(__shfl(int(incl), KMX_WAVE - 1));
    if (sh_out && total) {
        uint32_t at = valid + incl - mine;
#pragma unroll
        for (int j = 0; j < E; ++j)
            if ((res >> j) & 1u) sh_out[at++] = a[j] - delta;
    }
    wsync();                                                           // the stage is the next tile's / query's
    return valid + total;
}

__device__ __forceinline__ void validate_wide_wave(const uint32_t* __restrict__ arena, const QueryDesc& d, uint32_t q,
                                                   uint64_t* __restrict__ mask_words, uint32_t* __restrict__ flat)
{
    const uint32_t lane = lane_id();
    const uint32_t c0 = uint32_t(__builtin_amdgcn_readfirstlane(int(d.c0[q])));
    const uint64_t src = d.src[q] & ~SRC_FLAGS;
    const uint64_t p1 = d.p1[q], p1src = d.key[q], wbase = d.aux[q];
    const uint32_t pcnt = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(p1)))), delta = uint32_t(p1 >> 32) & KMX_P1_DELTA_MASK;
    uint32_t P = 512;                                                   // pcnt in (256, KMX_VWIDE]
    while (P < pcnt) P <<= 1;
    uint64_t* __restrict__ words = mask_words + wbase;
    uint32_t* __restrict__ sh_out = d.stitch_hits ? d.stitch_hits + wbase * 64 : nullptr;
    const uint32_t* __restrict__ cand = arena + src;
    const uint32_t* __restrict__ fil = arena + p1src;
    uint32_t valid = 0;
    for (uint32_t t0 = 0; t0 < c0; t0 += KMX_VWIDE_TILE) {
        const uint32_t n_c = min(uint32_t(KMX_VWIDE_TILE), c0 - t0);
        uint64_t* __restrict__ w = words + t0 / 64;
        if (n_c <= 512) valid = wide_tile<8>(cand + t0, n_c, fil, pcnt, P, delta, w, sh_out, valid, flat);
        else if (n_c <= 1024) valid = wide_tile<16>(cand + t0, n_c, fil, pcnt, P, delta, w, sh_out, valid, flat);
        else valid = wide_tile<32>(cand + t0, n_c, fil, pcnt, P, delta, w, sh_out, valid, flat);
    }
    if (lane == 0) {
        if ((c0 & 63) == 0) words[c0 / 64] = 0;                         // n_bits/64 + 1 words (compressed_bitset.hpp:23)
        d.cnt[q] = valid;
    }
}

#define KMX_VWIDE_BLOCK 256                 // four waves per block: 4 x 8 KB of stage, five blocks per CU
__global__ __launch_bounds__(KMX_VWIDE_BLOCK, 4) void k_validate_wide(const uint32_t* __restrict__ arena, QueryDesc d, uint64_t n_stitch,
                                                                   uint64_t* __restrict__ mask_words)
{
    __shared__ __attribute__((aligned(16))) uint32_t stage[KMX_VWIDE_BLOCK / KMX_WAVE][KMX_VWIDE + KMX_VWIDE_PAD];
    const uint32_t lane = lane_id();
    const uint64_t wave = (uint64_t(blockIdx.x) * KMX_VWIDE_BLOCK + threadIdx.x) / KMX_WAVE;
    const uint64_t n_waves = uint64_t(gridDim.x) * (KMX_VWIDE_BLOCK / KMX_WAVE);
    // few list entries per wave and round, so that the flagged queries spread over many waves
    for (uint64_t i0 = wave * KMX_VWIDE_SCAN; i0 < n_stitch; i0 += n_waves * KMX_VWIDE_SCAN) {
        const uint64_t i = i0 + lane;
        const bool have = lane < KMX_VWIDE_SCAN && i < n_stitch;
        const uint32_t q = have ? d.stitch_list[i] : 0u;
        const uint64_t p1 = have ? d.p1[q] : 0;
        uint64_t todo = __ballot(have && stitch_is_wide(p1));
        while (todo) {
            const int l = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            validate_wide_wave(arena, d, uint32_t(__shfl(int(q), l)), mask_words, stage[threadIdx.x / KMX_WAVE]);
        }
    }
}

// k_validate_tiny — STITCH queries whose SHORTER bucket (first part's or filter part's) has at most KMX_VTINY entries and whose
// longer one at most 2 KMX_VTINY (the regime of a large k: buckets of a few positions, e.g. 24-letter reads as two 12-mers):
// one THREAD per query does everything — both buckets with six straight-line 16-byte loads (one memory round trip), the filter
// as 8 x 16 register compares (the shorter list against the longer, whichever part it is: membership is symmetric), the
// further parts one after the other for the few survivors, the one mask word, the survivor list, the count.  A 16-lane group
// per query would idle 15 lanes there.
// direct: a query whose survivors are ONE RUN of its first bucket (or none) leaves as a plain copy of that run — its src is
// redirected there and k_fill treats it like an exact lookup (not when the candidate run is the caller's: KEEP_MASKS).
__global__ __launch_bounds__(KMX_BLOCK) void k_validate_tiny(const KmxIndexDev* __restrict__ ix,
                                                             const uint32_t* __restrict__ arena,
                                                             const uint8_t* __restrict__ qranks,
                                                             const uint64_t* __restrict__ qoff, QueryDesc d,
                                                             const uint32_t* __restrict__ list, uint64_t n_tiny,
                                                             uint64_t* __restrict__ mask_words, bool direct)
{
    const uint64_t i = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    if (i >= n_tiny) return;
    const uint32_t q = list[i];
    const uint32_t c0 = d.c0[q];
    const uint64_t src = d.src[q] & ~SRC_FLAGS;
    const uint64_t p1 = d.p1[q], p1src = d.key[q], wbase = d.aux[q];
    const bool more = (p1 & KMX_P1_MORE) != 0;
    const uint32_t pcnt = uint32_t(p1), delta = uint32_t(p1 >> 32) & KMX_P1_DELTA_MASK;
    static_assert(KMX_VTINY == 8, "two 16-byte loads for the shorter bucket, four for the longer");
    // (any element of the arena may be read 64 bytes wide: its allocation is padded)
    const KMX_GLOBAL uint32_t* ar = as_global(arena);
    // S: the shorter list (<= 8 entries), L: the longer (<= 16).  first_is_short: S is the first part's bucket — an entry s of S
    // meets the entry s + delta of L; otherwise S is the filter part's and s - delta is looked for (s < delta wraps past every
    // position: n + k - 1 < 2^32, kmer_index.hpp:169-170)
    const bool first_is_short = c0 <= pcnt;
    const uint64_t s_at = first_is_short ? src : p1src, l_at = first_is_short ? p1src : src;
    const uint32_t n_s = first_is_short ? c0 : pcnt, n_l = first_is_short ? pcnt : c0;
    const uint32_t shift = first_is_short ? delta : 0u - delta;
    const u32x4 s0 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + s_at), s1 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + s_at + 4);
    const u32x4 l0 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + l_at), l1 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + l_at + 4);
    const u32x4 l2 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + l_at + 8), l3 = *reinterpret_cast<const KMX_GLOBAL u32x4_a4*>(ar + l_at + 12);
    const uint32_t sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const uint32_t lv[16] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2.x, l2.y, l2.z, l2.w, l3.x, l3.y, l3.z, l3.w};
    // bit i of alive <-> candidate i of the FIRST part's bucket survives
    uint32_t alive = 0;
#pragma unroll
    for (uint32_t a = 0; a < 8; ++a) {
        uint32_t where = 0;                                                                        // bit b: S[a] meets L[b]
#pragma unroll
        for (uint32_t b = 0; b < 16; ++b) where |= uint32_t(sv[a] + shift == lv[b]) << b;           // binary_search :283, lower_bound :544-546
        where &= (1u << n_l) - 1u;
        if (a < n_s) alive |= first_is_short ? uint32_t(where != 0) << a : where;
    }
    if (more && alive) {
#pragma unroll 1
        for (uint32_t left = alive; left; left &= left - 1) {
            const uint32_t ci = uint32_t(__ffs(int(left))) - 1u;
            if (!stitch_parts_hold(ix, arena, qranks, qoff, q, arena[src + ci], 0u, 1u)) alive &= ~(1u << ci);
        }
    }
    const uint32_t valid = uint32_t(__popc(alive));
    mask_words[wbase] = alive;                                                                     // c0 <= 16 < 64: one word (compressed_bitset.hpp:23); bit i = word i>>6, bit i&63
    d.cnt[q] = valid;
    const uint32_t lo = alive ? uint32_t(__ffs(int(alive))) - 1u : 0u;
    if (direct && (alive >> lo) == (1u << valid) - 1u) {
        d.src[q] = src + lo;                                                                       // one run of the bucket (or nothing): copied where it lies
    } else if (d.stitch_hits) {
        uint32_t at = 0;
#pragma unroll 1
        for (uint32_t left = alive; left; left &= left - 1) d.stitch_hits[wbase * 64 + at++] = arena[src + uint32_t(__ffs(int(left))) - 1u];
    }
}

// A long survivor list (a query inside a repeat of the text: thousands of candidates pass the filter) taken by the whole
// wave: lane = survivor, two per lane, the parts one after the other — each lane computes the bucket of one part, the
// wave then walks the 64 buckets, every lane binary-searching its survivors in lockstep — so that 128 searches are in
// flight where a 16-lane group would run one survivor at a time.  Compaction in place, dropped survivors lose their
// mask bit (several lanes may hit one word: atomic).
#define KMX_VMORE_WAVE 48      // survivor lists longer than this go to the whole wave
#define KMX_VWAVE_SCAN 8       // k_validate_wave: list entries a wave looks at per round
__device__ void validate_more_wave(const KmxIndexDev* __restrict__ ix, const uint32_t* __restrict__ arena,
                                   const uint8_t* __restrict__ qranks, const uint64_t* __restrict__ qoff, const QueryDesc& d,
                                   uint32_t q, uint32_t tentative, uint64_t* __restrict__ mask_words)
{
    const uint32_t lane = lane_id();
    const uint64_t qb = qoff[q];
    const uint64_t m = qoff[q + 1] - qb;
    const uint8_t* __restrict__ qr = qranks + qb;
    const uint8_t* __restrict__ qend = qranks + qoff[q + 1];
    const uint32_t sigma = ix->sigma;
    const KmxPlanEntry pe = load_plan(ix, m);
    const bool single = pe.scheme == KMX_SCHEME_SINGLE;
    const KmxElemDev* __restrict__ sel = single ? &ix->elems[pe.elem] : nullptr;
    const uint32_t sk = single ? sel->k : 0u;
    const uint32_t sP = single ? uint32_t(m / sk) : 0u;
    const uint32_t n_extra = single ? sP - 1 + ((m % sk) ? 1u : 0u) : pe.nparts - 1u;
    const uint64_t wbase = d.aux[q];
    uint32_t* __restrict__ hits = d.stitch_hits + wbase * 64;
    const uint64_t src = d.src[q] & ~SRC_FLAGS;
    const uint32_t c0 = d.c0[q];
    const uint64_t below = (uint64_t(1) << lane) - 1;
    uint32_t kept = 0;
    for (uint32_t base = 0; base < tentative; base += 2 * KMX_WAVE) {
        const bool v0 = base + lane < tentative, v1 = base + KMX_WAVE + lane < tentative;
        const uint32_t p0 = v0 ? hits[base + lane] : 0u, p1 = v1 ? hits[base + KMX_WAVE + lane] : 0u;
        bool a0 = v0, a1 = v1;
        for (uint32_t pb = 0; pb < n_extra && __any(a0 || a1); pb += KMX_WAVE) {
            // lane = part pb + lane: its bucket and its offset in the query (as stitch_parts_hold)
            const uint32_t e = pb + lane;
            uint64_t rsrc = 0;
            uint32_t rcnt = 0, rdl = 0;
            if (e < n_extra) {
                const KmxElemDev* __restrict__ el = sel;
                uint32_t k = sk;
                uint64_t start;
                if (single) {
                    start = (e < sP - 1) ? uint64_t(e + 1) * sk : (m - sk);
                } else {
                    uint64_t mm = m;
                    for (uint32_t w = 0; w <= e; ++w) {
                        const KmxPlanEntry en = load_plan_chain(ix, mm);
                        el = &ix->elems[en.elem];
                        k = el->k;
                        mm -= k;
                    }
                    start = mm;
                }
                uint64_t h;
                rank_hash(qr + start, k, sigma, h, qend);
                const Run r = probe(el, h);
                rsrc = r.src; rcnt = r.cnt; rdl = uint32_t(start);
            }
            const uint32_t nb = min(uint32_t(KMX_WAVE), n_extra - pb);
            for (uint32_t j = 0; j < nb && __any(a0 || a1); ++j) {
                const uint64_t bs = __shfl(rsrc, int(j));
                const uint32_t bn = uint32_t(__shfl(int(rcnt), int(j)));
                const uint32_t dl = uint32_t(__shfl(int(rdl), int(j)));
                const uint32_t* __restrict__ bk = arena + bs;
                const uint32_t x0 = p0 + dl, x1 = p1 + dl;
                uint32_t c0s = 0, c1s = 0;                                  // branch-free halving, both searches in lockstep
                uint64_t P2 = 1;
                while (P2 <= bn) P2 <<= 1;
                for (uint32_t st = uint32_t(P2 >> 1); st; st >>= 1) {
                    const uint32_t i0 = min(c0s + st - 1, bn ? bn - 1 : 0u), i1 = min(c1s + st - 1, bn ? bn - 1 : 0u);
                    const uint32_t t0 = bk[i0], t1 = bk[i1];
                    c0s += (c0s + st - 1 < bn && t0 < x0) ? st : 0u;
                    c1s += (c1s + st - 1 < bn && t1 < x1) ? st : 0u;
                }
                a0 = a0 && c0s < bn && bk[min(c0s, bn ? bn - 1 : 0u)] == x0;   // binary_search :283, lower_bound :544-546
                a1 = a1 && c1s < bn && bk[min(c1s, bn ? bn - 1 : 0u)] == x1;
            }
        }
        // dropped survivors: their index among the candidates is their rank in the first part's bucket
        if (v0 && !a0) {
            const uint64_t idx = lower_bound_dev<uint32_t>(arena + src, c0, p0);
            atomicAnd(reinterpret_cast<unsigned long long*>(mask_words + wbase + (idx >> 6)), ~(1ull << (idx & 63)));
        }
        if (v1 && !a1) {
            const uint64_t idx = lower_bound_dev<uint32_t>(arena + src, c0, p1);
            atomicAnd(reinterpret_cast<unsigned long long*>(mask_words + wbase + (idx >> 6)), ~(1ull << (idx & 63)));
        }
        // compaction in place: everything this round read lies at or behind `base`, everything it writes before base + 128
        const uint64_t b0 = __ballot(a0), b1 = __ballot(a1);
        if (a0) hits[kept + uint32_t(__popcll(b0 & below))] = p0;
        kept += uint32_t(__popcll(b0));
        if (a1) hits[kept + uint32_t(__popcll(b1 & below))] = p1;
        kept += uint32_t(__popcll(b1));
    }
    if (lane == 0) d.cnt[q] = kept;
}

// A STITCH query whose first part has a long bucket (KMX_P1_BIG: it starts inside a repeat of the text), by the whole
// wave and without the filter stage: every part's bucket is looked up (lane = part), the SMALLEST one becomes the
// anchor — a query that straddles the repeat is anchored outside it — and its entries a, as start positions p = a - offset,
// are checked against all other parts, lane = candidate, two per lane, in lockstep.  The survivors go to stitch_hits
// (ascending, as the anchor's bucket is) and the count to cnt; the mask words of such a query are not produced (nobody
// reads them without KEEP_MASKS, and with it the query is not flagged).
#define KMX_BIG_TILE 192       // anchor entries per tile (three per lane): a part as dense as the anchor then needs a window of 256
#define KMX_BIG_STAGE 1024     // entries of a part's slice staged in LDS per wave
__device__ void validate_big_wave(const KmxIndexDev* __restrict__ ix, const uint32_t* __restrict__ arena,
                                  const uint8_t* __restrict__ qranks, const uint64_t* __restrict__ qoff, const QueryDesc& d,
                                  uint32_t q, uint32_t* __restrict__ wstage)
{
    const uint32_t lane = lane_id();
    const uint64_t qb = qoff[q];
    const uint64_t m = qoff[q + 1] - qb;
    const uint8_t* __restrict__ qr = qranks + qb;
    const uint8_t* __restrict__ qend = qranks + qoff[q + 1];
    const uint32_t sigma = ix->sigma;
    const KmxPlanEntry pe = load_plan(ix, m);
    const bool single = pe.scheme == KMX_SCHEME_SINGLE;
    const KmxElemDev* __restrict__ sel = single ? &ix->elems[pe.elem] : nullptr;
    const uint32_t sk = single ? sel->k : 0u;
    const uint32_t sP = single ? uint32_t(m / sk) : 0u;
    const uint32_t n_further = single ? sP - 1 + ((m % sk) ? 1u : 0u) : pe.nparts - 1u;
    const uint32_t n_all = n_further + 1;                              // index n_further: the first part (offset 0)
    // bucket and offset of part e (as stitch_parts_hold; e == n_further: the first part)
    auto part_of = [&](uint32_t e, uint64_t& rsrc, uint32_t& rcnt, uint32_t& rdl) {
        rsrc = 0; rcnt = 0; rdl = 0;
        if (e >= n_all) return;
        const KmxElemDev* __restrict__ el = sel;
        uint32_t k = sk;
        uint64_t start;
        if (single) {
            start = e == n_further ? 0 : (e < sP - 1) ? uint64_t(e + 1) * sk : (m - sk);
        } else {
            uint64_t mm = m;
            for (uint32_t w = 0; w <= e; ++w) {
                const KmxPlanEntry en = load_plan_chain(ix, mm);
                el = &ix->elems[en.elem];
                k = el->k;
                mm -= k;
            }
            start = mm;
        }
        uint64_t h;
        rank_hash(qr + start, k, sigma, h, qend);
        const Run r = probe(el, h);
        rsrc = r.src; rcnt = r.cnt; rdl = uint32_t(start);
    };
    // 1. the anchor: the part with the fewest positions
    uint64_t a_src = 0;
    uint32_t a_cnt = 0xFFFFFFFFu, a_dl = 0, a_e = 0;
    for (uint32_t pb = 0; pb < n_all; pb += KMX_WAVE) {
        uint64_t rsrc; uint32_t rcnt, rdl;
        part_of(pb + lane, rsrc, rcnt, rdl);
        uint64_t best = (uint64_t(pb + lane < n_all ? rcnt : 0xFFFFFFFFu) << 32) | (pb + lane);
        for (int off = 32; off > 0; off >>= 1) best = min(best, uint64_t(__shfl_xor(best, off)));
        if (uint32_t(best >> 32) < a_cnt) {
            const int owner = int(uint32_t(best) - pb);
            a_cnt = uint32_t(best >> 32); a_e = uint32_t(best);
            a_src = __shfl(rsrc, owner); a_dl = uint32_t(__shfl(int(rdl), owner));
        }
    }
    // 2. its entries against every other part, a tile of KMX_BIG_TILE entries at a time — a merge of sorted lists in tiles.
    //    Anchor and parts are ascending, so the start positions [p_lo, p_hi] of a tile can only be met by the entries of a part's
    //    bucket that lie in [p_lo + offset, p_hi + offset], and those follow the entries the previous tile looked at: every
    //    part keeps a cursor into its bucket (lane = part), the wave stages a window of the bucket from the cursor on in LDS
    //    with one coalesced load, searches the tile's candidates there (lockstep halving at LDS latency) and moves the
    //    cursor past the tile's range.  Linear in the lengths of the lists, one HBM round trip per tile and part, where a
    //    binary search per candidate and part over whole buckets (kmer_index.hpp:279-291) pays log2(bucket) dependent
    //    round trips each.  Queries with more than 64 parts keep no cursors: the range of a part is found by two binary
    //    searches per tile instead.
    uint32_t* __restrict__ hits = d.stitch_hits + d.aux[q] * 64;
    const uint64_t below = (uint64_t(1) << lane) - 1;
    constexpr int R = KMX_BIG_TILE / KMX_WAVE;
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const bool cursors = n_all <= KMX_WAVE;
    uint64_t c_src = 0;                                                   // this lane's part (cursor mode): bucket, size, offset, cursor
    uint32_t c_cnt = 0, c_dl = 0, c_cur = 0;
    uint32_t c_win = KMX_BIG_STAGE;                                       // window: what a tile is expected to need of this bucket (x 1.25), a power of two
    if (cursors) {
        part_of(lane, c_src, c_cnt, c_dl);
        const uint64_t want = (uint64_t(KMX_BIG_TILE) * c_cnt * 5) / (uint64_t(max(a_cnt, 1u)) * 4) + 8;
        c_win = 64;
        while (c_win < want && c_win < KMX_BIG_STAGE) c_win <<= 1;
        const uint32_t e_first = a_cnt ? arena[a_src] : 0u;
        c_cur = uint32_t(lower_bound_dev<uint32_t>(arena + c_src, c_cnt, (e_first >= a_dl ? e_first - a_dl : 0u) + c_dl));
    }
    uint32_t kept = 0;
    for (uint32_t base = 0; base < a_cnt; base += KMX_BIG_TILE) {
        uint32_t p[R];
        bool alive[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = base + uint32_t(r) * KMX_WAVE + lane;
            const uint32_t e = i < a_cnt ? arena[a_src + i] : 0u;
            alive[r] = i < a_cnt && e >= a_dl;                          // a start before the text is none
            p[r] = e - a_dl;
        }
        const uint32_t e_lo = arena[a_src + base], e_hi = arena[a_src + min(base + uint32_t(KMX_BIG_TILE), a_cnt) - 1];
        const uint32_t p_lo = e_lo >= a_dl ? e_lo - a_dl : 0u, p_hi = e_hi >= a_dl ? e_hi - a_dl : 0u;
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; ++r) any |= alive[r];
        for (uint32_t pb = 0; pb < n_all && __any(any); pb += KMX_WAVE) {
            uint64_t rsrc = c_src; uint32_t rcnt = c_cnt, rdl = c_dl, r0 = c_cur, r1 = c_cnt;
            if (!cursors) {
                part_of(pb + lane, rsrc, rcnt, rdl);
                r0 = uint32_t(lower_bound_dev<uint32_t>(arena + rsrc, rcnt, p_lo + rdl));
                r1 = uint32_t(upper_bound_dev<uint32_t>(arena + rsrc, rcnt, p_hi + rdl));
            }
            const uint32_t nb = min(uint32_t(KMX_WAVE), n_all - pb);
            // The first window of the NEXT part's bucket is asked for while this part's is searched (cursor mode: a part's cursor
            // only moves when the part itself is looked at, so the window is known a part ahead): the round per (tile, part) —
            // one global round trip, then a dozen dependent LDS steps — no longer waits for the memory system every time.
            constexpr uint32_t PRE_N = 4;                               // entries per lane of a prefetched window (8 measured slower: registers)
            constexpr uint32_t PRE = PRE_N * KMX_WAVE;
            auto next_part = [&](uint32_t j) { ++j; if (pb + j == a_e) ++j; return j; };
            auto window_of = [&](uint32_t j, uint32_t& cur, uint32_t& s1, uint32_t& win) {
                s1 = uint32_t(__shfl(int(r1), int(j))); cur = uint32_t(__shfl(int(r0), int(j))); win = uint32_t(__shfl(int(c_win), int(j)));
            };
            uint32_t pre[PRE_N];
#pragma unroll
            for (uint32_t u = 0; u < PRE_N; ++u) pre[u] = 0;
            uint32_t pre_j = 0xFFFFFFFFu;                               // the part whose first window sits in pre[]
            auto prefetch = [&](uint32_t j) {
                pre_j = 0xFFFFFFFFu;
                if (!cursors || j >= nb) return;
                uint32_t cur, s1, win;
                window_of(j, cur, s1, win);
                const uint32_t W = min(win, s1 > cur ? s1 - cur : 0u);
                if (W == 0 || W > PRE) return;
                const uint32_t* __restrict__ bkn = arena + __shfl(rsrc, int(j));
#pragma unroll
                for (uint32_t u = 0; u < PRE_N; ++u) pre[u] = bkn[cur + min(lane + u * KMX_WAVE, W - 1)];
                pre_j = j;
            };
            {
                uint32_t j0 = 0;
                if (pb + j0 == a_e) ++j0;
                prefetch(j0);
            }
            for (uint32_t j = 0; j < nb && __any(any); ++j) {
                if (pb + j == a_e) continue;                            // the anchor itself
                uint32_t s1, cur, win;
                window_of(j, cur, s1, win);
                const uint32_t dl = uint32_t(__shfl(int(rdl), int(j)));
                const uint32_t* __restrict__ bk = arena + __shfl(rsrc, int(j));
                const uint32_t x_hi = p_hi + dl;
                uint32_t und = 0;                                       // bit r: candidate r is alive and not yet decided for this part
                uint32_t x[R];
#pragma unroll
                for (int r = 0; r < R; ++r) { x[r] = p[r] + dl; und |= uint32_t(alive[r]) << r; }
                bool first_window = true;
                for (;;) {                                              // windows of the bucket (wave-uniform control)
                    const uint32_t W = min(win, s1 > cur ? s1 - cur : 0u);
                    if (W == 0) break;
                    uint32_t P2 = 1;
                    while (P2 < W) P2 <<= 1;
                    wsync();                                            // the previous window has been searched
                    if (first_window && pre_j == j) {
                        // (this window came with the previous part; now the next part's is asked for)
                        uint32_t mine[PRE_N];
#pragma unroll
                        for (uint32_t u = 0; u < PRE_N; ++u) mine[u] = pre[u];
                        prefetch(next_part(j));
#pragma unroll
                        for (uint32_t u = 0; u < PRE_N; ++u) {
                            const uint32_t t = lane + u * KMX_WAVE;
                            if (t < P2) wstage[t] = t < W ? mine[u] : 0xFFFFFFFFu;
                        }
                    } else {
                        if (first_window) prefetch(next_part(j));
                        for (uint32_t t = lane; t < P2; t += KMX_WAVE) wstage[t] = t < W ? bk[cur + t] : 0xFFFFFFFFu;
                    }
                    first_window = false;
                    wsync();
                    const uint32_t last_val = wstage[W - 1];
                    uint32_t pos[R + 1];
#pragma unroll
                    for (int r = 0; r <= R; ++r) pos[r] = 0;
                    for (uint32_t st = P2 >> 1; st; st >>= 1) {         // branch-free halving: the lane's candidates and the tile's end in lockstep
#pragma unroll
                        for (int r = 0; r < R; ++r) pos[r] += wstage[pos[r] + st - 1] < x[r] ? st : 0u;
                        pos[R] += wstage[pos[R] + st - 1] <= x_hi ? st : 0u;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const bool found = wstage[pos[r]] == x[r];      // binary_search :283, lower_bound :544-546
                        const bool inside = x[r] <= last_val;           // the window covers x: its verdict is final
                        if (((und >> r) & 1u) && (found || inside)) { und &= ~(1u << r); alive[r] = found; }
                    }
                    if (last_val >= x_hi || cur + W >= s1) {
                        cur += pos[R] + (wstage[pos[R]] <= x_hi ? 1u : 0u);   // entries <= the tile's last value: behind the cursor from now on
                        break;
                    }
                    cur += W;
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if ((und >> r) & 1u) alive[r] = false;              // the bucket ended before reaching p + offset
                if (cursors && lane == j) c_cur = cur;
                any = false;
#pragma unroll
                for (int r = 0; r < R; ++r) any |= alive[r];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t bal = __ballot(alive[r]);
            if (alive[r]) hits[kept + uint32_t(__popcll(bal & below))] = p[r];
            kept += uint32_t(__popcll(bal));
        }
    }
    if (lane == 0) d.cnt[q] = kept;
}

// k_validate_more_thread — the same check as k_validate_more with one THREAD per query, for queries with few further parts
// (reads a little longer than 2k: m = 25 at k = 10 has ONE part beyond the filter).  A 16-lane group per survivor has work
// for as many lanes as there are parts, and every part is a chain of dependent loads (letters, table, log2(bucket) probes):
// with 2 of 16 lanes busy the kernel is short of loads in flight, not of bandwidth.  Here every lane carries a query; the
// part k_validate filtered with is not looked at again.  A handled query loses its KMX_P1_MORE flag, which is what
// k_validate_more and k_validate_wave (launched behind this kernel for the queries it leaves) select by.
#define KMX_VMORE_THREAD_PARTS 4          // further parts up to which a query takes this path ...
#define KMX_VMORE_THREAD_PARTS_MANY 10    // ... and in a batch with at least KMX_VMORE_THREAD_MANY such queries (enough threads to
#define KMX_VMORE_THREAD_MANY (1u << 18)  //     hide a chain that is parts times as long: 100-letter reads at k = 10, 2e6 of them)
__global__ __launch_bounds__(KMX_BLOCK) void k_validate_more_thread(const KmxIndexDev* __restrict__ ix,
                                                                    const uint32_t* __restrict__ arena,
                                                                    const uint8_t* __restrict__ qranks,
                                                                    const uint64_t* __restrict__ qoff, QueryDesc d,
                                                                    uint64_t n_stitch, uint32_t max_parts,
                                                                    uint64_t* __restrict__ mask_words)
{
    const uint64_t i = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    if (i >= n_stitch) return;
    const uint32_t q = d.stitch_list[i];
    const uint64_t p1 = d.p1[q];
    if ((p1 & KMX_P1_BIG) || !(p1 & KMX_P1_MORE)) return;
    const uint32_t tentative = d.cnt[q];
    if (tentative == 0 || tentative > KMX_VMORE_WAVE) return;
    const uint64_t m = qoff[q + 1] - qoff[q];
    const KmxPlanEntry pe = load_plan(ix, m);
    uint32_t n_extra;
    if (pe.scheme == KMX_SCHEME_SINGLE) {
        const uint32_t k = ix->elems[pe.elem].k;
        n_extra = uint32_t(m / k) - 1u + ((m % k) ? 1u : 0u);
    } else {
        n_extra = pe.nparts - 1u;
    }
    if (n_extra > max_parts) return;
    const uint32_t delta = uint32_t(p1 >> 32) & KMX_P1_DELTA_MASK;
    const uint64_t wbase = d.aux[q];
    uint32_t* __restrict__ hits = d.stitch_hits + wbase * 64;
    uint32_t kept = 0;
    for (uint32_t t = 0; t < tentative; ++t) {
        const uint32_t p = hits[t];
        if (stitch_parts_hold(ix, arena, qranks, qoff, q, p, 0u, 1u, delta)) {
            if (kept != t) hits[kept] = p;
            ++kept;
        } else {
            // the candidates are ascending: the survivor's index is its rank in the first part's bucket
            const uint64_t src = d.src[q] & ~SRC_FLAGS;
            const uint64_t idx = lower_bound_dev<uint32_t>(arena + src, d.c0[q], p);
            mask_words[wbase + (idx >> 6)] &= ~(uint64_t(1) << (idx & 63));
        }
    }
    d.cnt[q] = kept;
    d.p1[q] = p1 & ~KMX_P1_MORE;
}

// k_validate_more — STITCH queries with further parts beyond the filter of k_validate<false>: every survivor
// the filter left in stitch_hits is checked against all parts (one part per lane of the query's group);
// the list is compacted in place, the mask bit of a dropped survivor is cleared and cnt is corrected.
__global__ __launch_bounds__(KMX_BLOCK) void k_validate_more(const KmxIndexDev* __restrict__ ix,
                                                             const uint32_t* __restrict__ arena,
                                                             const uint8_t* __restrict__ qranks,
                                                             const uint64_t* __restrict__ qoff, QueryDesc d,
                                                             uint64_t n_stitch, uint64_t* __restrict__ mask_words)
{
    const uint32_t lane = lane_id();
    const uint32_t g = lane / KMX_VGROUP, gl = lane % KMX_VGROUP;
    const uint64_t wave = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    const uint64_t n_waves = uint64_t(gridDim.x) * (KMX_BLOCK / KMX_WAVE);
    for (uint64_t i0 = wave * KMX_VGROUPS; i0 < n_stitch; i0 += n_waves * KMX_VGROUPS) {
        const uint64_t i = i0 + g;
        const bool have = i < n_stitch;
        const uint32_t q = have ? d.stitch_list[i] : 0u;
        const uint64_t p1 = have ? d.p1[q] : 0;
        const uint32_t tentative = have ? d.cnt[q] : 0u;
        // (big queries and long survivor lists: k_validate_wave)
        const bool more = !(p1 & KMX_P1_BIG) && (p1 & KMX_P1_MORE) != 0 && tentative != 0;   // group-uniform
        const bool big = more && tentative > KMX_VMORE_WAVE;
        const bool active = more && !big;
        uint32_t rounds = active ? tentative : 0u;
#pragma unroll
        for (int e = 1; e < KMX_VGROUPS; ++e) rounds = max(rounds, uint32_t(__shfl_xor(int(rounds), e * KMX_VGROUP)));
        rounds = uint32_t(__builtin_amdgcn_readfirstlane(int(rounds)));
        if (rounds == 0) continue;
        const uint64_t wbase = active ? d.aux[q] : 0;
        uint32_t* __restrict__ hits = d.stitch_hits + wbase * 64;
        uint32_t kept = 0;
        for (uint32_t t = 0; t < rounds; ++t) {
            if (active && t < tentative) {
                const uint32_t p = hits[t];
                const bool good = stitch_parts_hold(ix, arena, qranks, qoff, q, p, gl);
                const uint32_t verdicts = uint32_t(__ballot(good) >> (KMX_VGROUP * g)) & 0xFFFFu;
                if (verdicts == 0xFFFFu) {
                    if (gl == 0 && kept != t) hits[kept] = p;
                    ++kept;
                } else if (gl == 0) {
                    // the candidates are ascending: the survivor's index is its rank in the first part's bucket
                    const uint64_t src = d.src[q] & ~SRC_FLAGS;
                    const uint64_t idx = lower_bound_dev<uint32_t>(arena + src, d.c0[q], p);
                    mask_words[wbase + (idx >> 6)] &= ~(uint64_t(1) << (idx & 63));
                }
            }
        }
        if (active && gl == 0) d.cnt[q] = kept;
    }
}

// k_validate_wave — the STITCH queries that want a whole wave: big ones (validate_big_wave) and those whose filter left
// a long survivor list (validate_more_wave).  Lane = list entry while looking for them, then the wave takes the flagged
// queries one by one.  A kernel of its own so that the registers of the wave-wide paths do not cost the group paths of
// k_validate_more their occupancy.
#ifndef KMX_VWAVE_OCC
#define KMX_VWAVE_OCC 4        // waves per SIMD k_validate_wave is compiled for
#endif
__global__ __launch_bounds__(KMX_BLOCK, KMX_VWAVE_OCC) void k_validate_wave(const KmxIndexDev* __restrict__ ix,
                                                             const uint32_t* __restrict__ arena,
                                                             const uint8_t* __restrict__ qranks,
                                                             const uint64_t* __restrict__ qoff, QueryDesc d,
                                                             uint64_t n_stitch, uint64_t* __restrict__ mask_words)
{
    __shared__ uint32_t stage[KMX_BLOCK / KMX_WAVE][KMX_BIG_STAGE];
    const uint32_t lane = lane_id();
    const uint64_t wave = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    const uint64_t n_waves = uint64_t(gridDim.x) * (KMX_BLOCK / KMX_WAVE);
    // KMX_VWAVE_SCAN list entries per wave and round: few enough that the flagged queries spread over many waves
    for (uint64_t i0 = wave * KMX_VWAVE_SCAN; i0 < n_stitch; i0 += n_waves * KMX_VWAVE_SCAN) {
        const uint64_t i = i0 + lane;
        const bool have = lane < KMX_VWAVE_SCAN && i < n_stitch;
        const uint32_t q = have ? d.stitch_list[i] : 0u;
        const uint64_t p1 = have ? d.p1[q] : 0;
        const uint32_t tentative = have ? d.cnt[q] : 0u;
        const bool whole = (p1 & KMX_P1_BIG) != 0;
        const bool longlist = !whole && (p1 & KMX_P1_MORE) != 0 && tentative > KMX_VMORE_WAVE;
        uint64_t todo = __ballot(whole || longlist);
        while (todo) {
            const int l = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t ql = uint32_t(__shfl(int(q), l));
            if (__shfl(int(whole), l)) validate_big_wave(ix, arena, qranks, qoff, d, ql, stage[threadIdx.x / KMX_WAVE]);
            else validate_more_wave(ix, arena, qranks, qoff, d, ql, uint32_t(__shfl(int(tentative), l)), mask_words);
        }
    }
}

// ---------------------------------------------------------------------------
// exclusive scan u32 -> u64 (reduce, spine, downsweep)
// ---------------------------------------------------------------------------
#define KMX_SCAN_ITEMS 16
#define KMX_SCAN_TILE (KMX_BLOCK * KMX_SCAN_ITEMS)

__device__ __forceinline__ uint64_t block_exclusive_scan_u64(uint64_t v, uint64_t* total_out)
{
    __shared__ uint64_t wave_sum[KMX_BLOCK / KMX_WAVE];
    const uint32_t lane = lane_id(), w = threadIdx.x / KMX_WAVE;
    uint64_t inc = v;
    for (int off = 1; off < KMX_WAVE; off <<= 1) {
        uint64_t t = __shfl_up(inc, off);
        if (lane >= uint32_t(off)) inc += t;
    }
    if (lane == KMX_WAVE - 1) wave_sum[w] = inc;
    __syncthreads();
    uint64_t carry = 0, total = 0;
    for (uint32_t i = 0; i < KMX_BLOCK / KMX_WAVE; ++i) {
        uint64_t s = wave_sum[i];
        if (i < w) carry += s;
        total += s;
    }
    __syncthreads();
    if (total_out) *total_out = total;
    return carry + inc - v;
}

// ---------------------------------------------------------------------------
// k_small — a handful of queries, start to finish in ONE launch of ONE workgroup.
//
// The general pipeline costs a batch of one 80-100 us: two input copies, three to five launches, a counter read-back and
// four to eight result copies, each a host round trip; the reference answers the same call in about a microsecond
// (kmer_index.hpp:505-558 on a warm hash map).  Here the queries are read from, and the complete result — hit_off,
// positions (= to_vector(), kmer_index_result.hpp:244-260), status, kinds, and for cross-referenced queries the candidate
// run + compressed_bitset words (kmer_index_result.hpp:15-24) — is written to one page-locked host block by the kernel
// itself: one launch and one stream wait per call.  Thread = query for the lookups (lookup_query, shared with k_lookup),
// the workgroup together for validation, sort and copy.  Same results as the general path, bit for bit.  A batch this
// kernel is not made for (too many hits, long candidate lists) sets `fallback` and the host takes the general path.
// ---------------------------------------------------------------------------
// number of further parts of a cross-referenced query of length m (the parts stitch_parts_hold walks)
__device__ __forceinline__ uint32_t stitch_n_extra(const KmxIndexDev* __restrict__ ix, uint64_t m)
{
    const KmxPlanEntry pe = load_plan(ix, m);
    if (pe.scheme != KMX_SCHEME_SINGLE) return pe.nparts - 1u;
    const uint32_t k = ix->elems[pe.elem].k;
    return uint32_t(m / k) - 1u + ((m % k) ? 1u : 0u);
}

#define KMX_SMALL_XFLAG (1ull << 63)          // an exchange word that has been published: FLAG | hits << 32 | mask words
#define KMX_SMALL_SPIN_LIMIT (1u << 20)       // polls of one exchange word before the workgroup gives up (~0.1 s)
__global__ __launch_bounds__(KMX_BLOCK) void k_small(const KmxIndexDev* __restrict__ ix, const uint32_t* __restrict__ arena,
                                                     unsigned char* __restrict__ mailbox, KmxSmallLayout L, KmxSmallArgs args,
                                                     uint32_t nq_total, unsigned long long* __restrict__ xchg, uint32_t flags)
{
    // a workgroup = up to 256 consecutive queries: queries [blk * 256, ...) of the batch
    const uint32_t blk = blockIdx.x, qb = blk * KMX_SMALL_NQ;
    const uint32_t nq = args.nq[blk], n_letters = args.n_letters[blk];
    const unsigned char* __restrict__ in_area = mailbox + size_t(blk) * KMX_SMALL_IN_BYTES;
    constexpr uint32_t NW = KMX_BLOCK / KMX_WAVE;
    static_assert(NW * KMX_SMALL_WCAP == KMX_SMALL_SORT, "the waves share the sort buffer of the workgroup");
    __shared__ __attribute__((aligned(16))) unsigned char s_in[KMX_SMALL_IN_BYTES + 16];
    __shared__ KmxElemDev elems_s[KMX_MAX_KS];
    __shared__ uint64_t s_src[KMX_SMALL_NQ], s_aux[KMX_SMALL_NQ];
    __shared__ uint32_t s_cnt[KMX_SMALL_NQ], s_c0[KMX_SMALL_NQ], s_off[KMX_SMALL_NQ + 1];
    __shared__ uint8_t s_kind[KMX_SMALL_NQ], s_status[KMX_SMALL_NQ];
    __shared__ uint64_t s_words[KMX_SMALL_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t s_sort[KMX_SMALL_SORT];        // sort buffer; as bytes: the candidates' verdicts
    __shared__ uint16_t s_wslow[KMX_SMALL_WSLOW], s_bslow[KMX_SMALL_BSLOW];
    __shared__ uint32_t s_n_wslow, s_n_bslow, s_n_words, s_bad, s_valid, s_n_stitch, s_n_prefix, s_n_error, s_n_none;
    const uint32_t tid = threadIdx.x, lane = lane_id(), wv = threadIdx.x / KMX_WAVE;
    KmxSmallHeader* __restrict__ hdr = reinterpret_cast<KmxSmallHeader*>(mailbox + L.off_header) + blk;
    // a workgroup that gives up still publishes its exchange word: the workgroups behind it must not wait for ever
    auto decline = [&] {
        if (threadIdx.x == 0) {
            hdr->fallback = 1;
            if (xchg) __hip_atomic_store((KMX_GLOBAL unsigned long long*)xchg + blk, KMX_SMALL_XFLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // 0. the queries out of host memory (one PCIe round trip for the lot), the element descriptors out of HBM
    {
        const uint32_t in_bytes = (nq + 1) * 8 + n_letters;
        const uint4* __restrict__ src4 = reinterpret_cast<const uint4*>(in_area);
        uint4* dst4 = reinterpret_cast<uint4*>(s_in);
        for (uint32_t i = tid; i < (in_bytes + 15) / 16; i += KMX_BLOCK) dst4[i] = src4[i];
        const uint32_t n_words = ix->n_ks * uint32_t(sizeof(KmxElemDev) / 8);
        const uint64_t* __restrict__ srcw = reinterpret_cast<const uint64_t*>(ix->elems);
        uint64_t* dstw = reinterpret_cast<uint64_t*>(elems_s);
        for (uint32_t i = tid; i < n_words; i += KMX_BLOCK) dstw[i] = srcw[i];
        if (tid == 0) { s_n_wslow = s_n_bslow = s_n_words = s_bad = s_n_stitch = s_n_prefix = s_n_error = s_n_none = 0; }
    }
    __syncthreads();
    const uint64_t* s_qoff = reinterpret_cast<const uint64_t*>(s_in);
    const uint8_t* s_qr = s_in + (nq + 1) * 8;
    const uint8_t* s_qend = s_qr + n_letters;

    // 1. thread = query: the descriptor (KEEP_MASKS: every cross-referenced query goes through the validation below)
    {
        LookupOut lo;
        lo.src = lo.aux = lo.key = lo.p1 = 0; lo.cnt = lo.c0 = 0; lo.kind = KMX_KIND_NONE; lo.status = KMX_Q_OK; lo.resolved = false;
        if (tid < nq) {
            const uint64_t b = s_qoff[tid], m = s_qoff[tid + 1] - b;
            lookup_query(ix, elems_s, s_qr + b, m, s_qend, flags | KMX_SEARCH_KEEP_MASKS, lo);
            uint32_t size = 0, words = 0;                              // size != 0: a slow query with that many candidates / positions
            if (lo.kind == KMX_KIND_STITCH) {
                size = max(lo.c0, 1u);
                words = lo.c0 / 64 + 1;                                // compressed_bitset.hpp:23
                atomicAdd(&s_n_stitch, 1u);
            } else if (lo.kind == KMX_KIND_PREFIX) {
                const uint32_t len = lo.cnt - uint32_t(__popcll(lo.aux));
                if (lo.c0 > 1 && len > 1) size = len;                  // several runs: the slice wants sorting (kmer_index_result.hpp:258)
                atomicAdd(&s_n_prefix, 1u);
            }
            if (lo.status != KMX_Q_OK) atomicAdd(&s_n_error, 1u);
            else if (lo.kind == KMX_KIND_NONE) atomicAdd(&s_n_none, 1u);
            if (size > KMX_SMALL_SORT) atomicOr(&s_bad, 1u);
            else if (size) {
                if (size <= KMX_SMALL_WCAP) {
                    const uint32_t slot = atomicAdd(&s_n_wslow, 1u);
                    if (slot < KMX_SMALL_WSLOW) s_wslow[slot] = uint16_t(tid);
                } else {
                    const uint32_t slot = atomicAdd(&s_n_bslow, 1u);
                    if (slot < KMX_SMALL_BSLOW) s_bslow[slot] = uint16_t(tid);
                }
                if (words) lo.aux = atomicAdd(&s_n_words, words);      // STITCH: index of the first mask word
            }
        }
        s_src[tid] = lo.src; s_aux[tid] = lo.aux; s_cnt[tid] = lo.cnt; s_c0[tid] = lo.c0; s_kind[tid] = lo.kind; s_status[tid] = lo.status;
    }
    __syncthreads();
    if (s_bad || s_n_wslow > KMX_SMALL_WSLOW || s_n_bslow > KMX_SMALL_BSLOW) {
        decline();
        return;
    }
    // fewer small slow queries than waves: the whole workgroup takes them too, one after the other (a wave per query would
    // leave the other waves idle)
    const bool merge = s_n_wslow < NW;
    const uint32_t n_wslow = merge ? 0u : s_n_wslow, n_bslow = s_n_bslow + (merge ? s_n_wslow : 0u);
    auto bslow_at = [&](uint32_t si) -> uint32_t { return si < s_n_bslow ? s_bslow[si] : s_wslow[si - s_n_bslow]; };

    // 2. cross-referenced queries: candidate p of the first part's bucket survives when every further part holds p + its
    //    offset (kmer_index.hpp:279-291, :541-551).  One (candidate, part) pair per thread — a query of a few candidates and
    //    a few parts is checked in one round trip through the parts' buckets; the verdicts meet in a byte per candidate;
    //    64 candidates = one ballot = one compressed_bitset word.  `T` threads (a wave or the workgroup) per query.
    auto validate = [&](uint32_t q, uint32_t t, uint32_t T, uint8_t* ok8, auto sync) -> uint32_t {
        const uint64_t src = s_src[q] & ~SRC_FLAGS;
        const uint32_t c0 = s_c0[q], wbase = uint32_t(s_aux[q]);
        const uint32_t n_extra = stitch_n_extra(ix, s_qoff[q + 1] - s_qoff[q]);
        for (uint32_t c = t; c < c0; c += T) ok8[c] = 1;
        sync();
        for (uint32_t item = t; item < c0 * n_extra; item += T) {
            const uint32_t c = item / n_extra, e = item - c * n_extra;
            if (!stitch_parts_hold(ix, arena, s_qr, s_qoff, q, arena[src + c], e, n_extra)) ok8[c] = 0;
        }
        sync();
        uint32_t mine = 0;
        for (uint32_t c = t; c < ((c0 + 64) & ~63u); c += T) {          // (c0 + 64) & ~63: covers word c0 / 64 even when c0 % 64 == 0
            const uint64_t bal = __ballot(c < c0 && ok8[c]);
            if (lane == 0) { s_words[wbase + c / 64] = bal; mine += uint32_t(__popcll(bal)); }
        }
        return mine;                                                    // (lane 0 of every wave: its share of the survivors)
    };
    for (uint32_t si = wv; si < n_wslow; si += NW) {                    // a wave per query
        const uint32_t q = s_wslow[si];
        if (s_kind[q] != KMX_KIND_STITCH) continue;
        const uint32_t mine = validate(q, lane, KMX_WAVE, reinterpret_cast<uint8_t*>(s_sort + wv * KMX_SMALL_WCAP), wsync);
        if (lane == 0) s_cnt[q] = mine;
        wsync();
    }
    __syncthreads();
    for (uint32_t si = 0; si < n_bslow; ++si) {                         // the workgroup per query
        const uint32_t q = bslow_at(si);
        if (s_kind[q] != KMX_KIND_STITCH) continue;                     // workgroup-uniform
        if (tid == 0) s_valid = 0;
        __syncthreads();
        const uint32_t mine = validate(q, tid, KMX_BLOCK, reinterpret_cast<uint8_t*>(s_sort), [] { __syncthreads(); });
        if (lane == 0 && mine) atomicAdd(&s_valid, mine);
        __syncthreads();
        if (tid == 0) s_cnt[q] = s_valid;
        __syncthreads();
    }

    // 3. offsets of the hit lists
    {
        uint64_t total = 0;
        const uint64_t ex = block_exclusive_scan_u64(tid < nq ? s_cnt[tid] : 0u, &total);
        if (total > KMX_SMALL_POS) {
            decline();
            return;
        }
        s_off[tid] = uint32_t(ex);
        if (tid == 0) s_off[KMX_SMALL_NQ] = uint32_t(total);
    }
    __syncthreads();
    const uint32_t total = s_off[KMX_SMALL_NQ];

    // 3b. where this workgroup's hits and mask words go in the batch's arrays: behind those of the workgroups in front.
    //     Every workgroup publishes one word (its totals) and reads the words of its predecessors — at most 31, all resident
    //     (the grid is at most 32 workgroups), polled by one wave with a bounded spin.
    uint32_t hit_base = 0, word_base = 0;
    if (xchg) {
        __shared__ uint32_t s_hit_base, s_word_base, s_gave_up;
        if (wv == 0) {
            if (lane == 0)
                __hip_atomic_store((KMX_GLOBAL unsigned long long*)xchg + blk, KMX_SMALL_XFLAG | (uint64_t(total) << 32) | s_n_words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long wd = KMX_SMALL_XFLAG;
            bool gave_up = false;
            if (lane < blk) {
                unsigned int spins = 0;
                for (;;) {
                    wd = __hip_atomic_load((KMX_GLOBAL unsigned long long*)xchg + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (wd & KMX_SMALL_XFLAG) break;
                    if (++spins > KMX_SMALL_SPIN_LIMIT) { gave_up = true; wd = KMX_SMALL_XFLAG; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            uint64_t hsum = (wd >> 32) & 0x7FFFFFFFu, wsum = wd & 0xFFFFFFFFu;
            for (int off = 32; off > 0; off >>= 1) { hsum += __shfl_xor(hsum, off); wsum += __shfl_xor(wsum, off); }
            if (lane == 0) { s_hit_base = uint32_t(hsum); s_word_base = uint32_t(wsum); s_gave_up = __any(gave_up) ? 1u : 0u; }
        }
        __syncthreads();
        if (s_gave_up) {
            if (tid == 0) hdr->fallback = 1;                            // (its own word is out already)
            return;
        }
        hit_base = s_hit_base; word_base = s_word_base;
    }

    // 4. the per-query arrays and the mask words, straight into the host block
    {
        uint64_t* __restrict__ o_off = reinterpret_cast<uint64_t*>(mailbox + L.off_hitoff) + qb;
        uint64_t* __restrict__ o_csrc = reinterpret_cast<uint64_t*>(mailbox + L.off_csrc) + qb;
        uint64_t* __restrict__ o_mbase = reinterpret_cast<uint64_t*>(mailbox + L.off_mbase) + qb;
        uint32_t* __restrict__ o_ccnt = reinterpret_cast<uint32_t*>(mailbox + L.off_ccnt) + qb;
        uint64_t* __restrict__ o_words = reinterpret_cast<uint64_t*>(mailbox + L.off_words) + word_base;
        if (tid < nq) {
            o_off[tid] = uint64_t(hit_base) + s_off[tid];
            o_csrc[tid] = s_src[tid] & ~SRC_FLAGS;
            o_mbase[tid] = s_aux[tid] + word_base;
            o_ccnt[tid] = s_c0[tid];
            mailbox[L.off_status + qb + tid] = s_status[tid];
            mailbox[L.off_kinds + qb + tid] = s_kind[tid];
        }
        if (tid == 0 && qb + nq == nq_total) o_off[nq] = uint64_t(hit_base) + total;     // (the last workgroup)
        for (uint32_t w = tid; w < s_n_words; w += KMX_BLOCK) o_words[w] = s_words[w];
    }
    uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(mailbox + L.off_pos) + hit_base;

    // 5a. plain copies, slot-centric: every output slot finds its query (the last one whose offset is <= the slot) and
    //     copies its element; the slots of slow queries are left to 5b
    //     A thread takes 16 CONSECUTIVE slots of a 4096-slot pass: one search for the first, then it walks the queries;
    //     its 16 loads are independent (all in flight together); the pass is staged in LDS and leaves for the host coalesced.
    //     (A handful of hits — at most four slots per thread: every slot finds its own query, nothing is staged.)
    auto slot_value = [&](uint32_t q, uint32_t idx) -> uint32_t {
        const uint8_t kind = s_kind[q];
        if (kind == KMX_KIND_EXACT) return arena[s_src[q] + idx];
        if (kind != KMX_KIND_PREFIX) return 0xFFFFFFFFu;                // (never a position: "nothing to store" — a slow query's slot)
        // the slice of every k-mer with this prefix, then the last-kmer offsets (kmer_index.hpp:138-146): bit j of aux <-> n - j
        const uint64_t tmask = s_aux[q];
        const uint32_t len = s_cnt[q] - uint32_t(__popcll(tmask));
        if (idx < len) return (s_c0[q] <= 1 || len <= 1) ? arena[(s_src[q] & ~SRC_FLAGS) + idx] : 0xFFFFFFFFu;     // one run: already ascending
        uint32_t t = idx - len;                                         // t-th smallest position = t-th highest bit
        uint64_t mm = tmask;
        int bit = 63 - __clzll(mm);
        while (t--) { mm &= ~(uint64_t(1) << bit); bit = 63 - __clzll(mm); }
        return uint32_t(ix->n - uint64_t(bit));
    };
    if (total <= 4 * KMX_BLOCK) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t s = tid + uint32_t(u) * KMX_BLOCK;
            v[u] = 0xFFFFFFFFu;
            if (s < total) {
                uint32_t lo = 0, hi = nq;                               // last q in [0, nq) with s_off[q] <= s
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_off[mid] <= s) lo = mid; else hi = mid; }
                v[u] = slot_value(lo, s - s_off[lo]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (v[u] != 0xFFFFFFFFu) out[tid + uint32_t(u) * KMX_BLOCK] = v[u];
    } else {
        constexpr uint32_t PASS = KMX_SMALL_SORT, PER = PASS / KMX_BLOCK;
        for (uint32_t p0 = 0; p0 < total; p0 += PASS) {
            const uint32_t s_first = p0 + tid * PER;
            uint32_t q = 0;
            if (s_first < total) {
                uint32_t lo = 0, hi = nq;                               // last q in [0, nq) with s_off[q] <= s_first
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_off[mid] <= s_first) lo = mid; else hi = mid; }
                q = lo;
            }
            uint32_t v[PER];
#pragma unroll
            for (uint32_t u = 0; u < PER; ++u) {
                const uint32_t s = s_first + u;
                v[u] = 0xFFFFFFFFu;                                     // (never a position: "nothing to store")
                if (s < total) {
                    while (q + 1 < nq && s_off[q + 1] <= s) ++q;        // the owner of slot s (queries without hits are stepped over)
                    v[u] = slot_value(q, s - s_off[q]);
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < PER; ++u) s_sort[tid * PER + u] = v[u];
            __syncthreads();
            const uint32_t n_here = min(PASS, total - p0);
            for (uint32_t i = tid; i < n_here; i += KMX_BLOCK) {
                const uint32_t x = s_sort[i];
                if (x != 0xFFFFFFFFu) out[p0 + i] = x;
            }
            __syncthreads();
        }
    }
    // 5b. the slow ones.  STITCH: decode the mask words — the survivor of candidate c lands at the number of set bits before
    //     it (is_valid + push_back, kmer_index_result.hpp:250-256; the candidates are ascending, so is the list).
    //     PREFIX: the slice sorted in LDS (the std::sort of kmer_index_result.hpp:258).
    auto emit = [&](uint32_t q, uint32_t t, uint32_t T, uint32_t* sbuf, auto sync) {
        const uint64_t src = s_src[q] & ~SRC_FLAGS;
        if (s_kind[q] == KMX_KIND_STITCH) {
            const uint32_t c0 = s_c0[q], wbase = uint32_t(s_aux[q]);
            for (uint32_t c = t; c < c0; c += T) {
                const uint64_t w = s_words[wbase + c / 64];
                if ((w >> (c & 63)) & 1u) {
                    uint32_t rank = uint32_t(__popcll(w & ((uint64_t(1) << (c & 63)) - 1)));
                    for (uint32_t j = 0; j < c / 64; ++j) rank += uint32_t(__popcll(s_words[wbase + j]));
                    out[s_off[q] + rank] = arena[src + c];
                }
            }
        } else {
            const uint32_t len = s_cnt[q] - uint32_t(__popcll(s_aux[q]));
            uint32_t n2 = 2;
            while (n2 < len) n2 <<= 1;
            sync();                                                     // sbuf is free again
            for (uint32_t i = t; i < n2; i += T) sbuf[i] = i < len ? arena[src + i] : 0xFFFFFFFFu;
            sync();
            bitonic_lds(sbuf, n2, t, T, sync);
            for (uint32_t i = t; i < len; i += T) out[s_off[q] + i] = sbuf[i];
        }
    };
    __syncthreads();                                                    // (the verdict bytes of step 2 lived in s_sort)
    for (uint32_t si = wv; si < n_wslow; si += NW) emit(s_wslow[si], lane, KMX_WAVE, s_sort + wv * KMX_SMALL_WCAP, wsync);
    __syncthreads();
    for (uint32_t si = 0; si < n_bslow; ++si) emit(bslow_at(si), tid, KMX_BLOCK, s_sort, [] { __syncthreads(); });
    if (tid == 0) {
        hdr->nq = nq;
        hdr->n_hits = total;
        hdr->n_mask_words = s_n_words;
        hdr->n_stitch = s_n_stitch; hdr->n_prefix = s_n_prefix; hdr->n_error = s_n_error; hdr->n_none = s_n_none;
        hdr->fallback = 0;
    }
}

void launch_small(hipStream_t s, const KmxIndexDev* ix, const uint32_t* arena, unsigned char* mailbox, const KmxSmallLayout& layout, uint32_t n_blocks,
                  const KmxSmallArgs& args, uint32_t nq_total, unsigned long long* xchg, uint32_t flags)
{
    hipLaunchKernelGGL(k_small, dim3(n_blocks), dim3(KMX_BLOCK), 0, s, ix, arena, mailbox, layout, args, nq_total, xchg, flags);
}

__global__ __launch_bounds__(KMX_BLOCK) void k_scan_reduce(const uint32_t* __restrict__ in, uint64_t n,
                                                           uint64_t* __restrict__ bsum)
{
    const uint64_t base = uint64_t(blockIdx.x) * KMX_SCAN_TILE;
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < KMX_SCAN_ITEMS; ++j) {
        uint64_t i = base + uint64_t(j) * KMX_BLOCK + threadIdx.x;
        if (i < n) s += in[i];
    }
    uint64_t total;
    block_exclusive_scan_u64(s, &total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(KMX_BLOCK) void k_scan_spine(uint64_t* __restrict__ bsum, uint64_t nb,
                                                          unsigned long long* __restrict__ total_out)
{
    uint64_t running = 0;
    for (uint64_t base = 0; base < nb; base += KMX_BLOCK) {
        uint64_t i = base + threadIdx.x;
        uint64_t v = i < nb ? bsum[i] : 0;
        uint64_t total;
        uint64_t ex = block_exclusive_scan_u64(v, &total);
        if (i < nb) bsum[i] = running + ex;
        running += total;
    }
    if (threadIdx.x == 0) *total_out = running;
}

// Downsweep.  With TILES it also does the work of k_partition for the offsets it produces: query i
// owns output slots [ex, ex + v); every tile boundary t*tile inside that range gets tile_q[t] = i
// (the last query whose offset is <= the boundary), and tile_q[t] for boundaries at or past the
// total is n - 1.
//
// SPINE: there is no spine launch — bsum holds the plain block sums of k_scan_reduce and every block adds up the
// sums in front of it itself (at most KMX_SCAN_FUSED_SPINE_BLOCKS of them: a few KB out of L2); the last block
// also writes the grand total.
// SPINE == 2: the same with the sums k_lookup left behind, one per KMX_BLOCK * (queries per thread of the variant that ran)
// queries (n_fine of them, fine_per_tile = KMX_SCAN_TILE / that per scan tile) — no reduce launch either.
template <bool TILES, int SPINE>
__global__ __launch_bounds__(KMX_BLOCK) void k_scan_down(const uint32_t* __restrict__ in, uint64_t n,
                                                         const uint64_t* __restrict__ bsum,
                                                         uint64_t* __restrict__ out, uint64_t tile,
                                                         uint64_t n_tiles_cap, uint32_t* __restrict__ tile_q,
                                                         unsigned long long* __restrict__ total_out, uint64_t n_fine,
                                                         CounterPub pub, uint32_t fine_per_tile)
{
    static_assert(KMX_SCAN_TILE % (KMX_BLOCK * 8) == 0 && KMX_SCAN_TILE % (KMX_BLOCK * 4) == 0, "a scan tile is a whole number of lookup blocks");
    // The block's 4096 items as KMX_SCAN_ROWS rows of 1024: in row r thread t owns items r*1024 + 4t .. 4t+3 —
    // one coalesced 16-byte load and two 16-byte stores per row.  The rows are scanned together (one set of
    // wave shuffles and barriers for all of them).
    constexpr int ROWS = KMX_SCAN_ITEMS / 4;
    __shared__ uint64_t wave_sum[ROWS][KMX_BLOCK / KMX_WAVE];
    const uint32_t lane = lane_id(), w = threadIdx.x / KMX_WAVE;
    const uint64_t block_base = uint64_t(blockIdx.x) * KMX_SCAN_TILE;
    uint32_t v[ROWS][4];
    uint64_t inc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint64_t i = block_base + uint64_t(r) * (KMX_BLOCK * 4) + uint64_t(threadIdx.x) * 4;
        if (i + 3 < n) {
            const uint4 t = *reinterpret_cast<const uint4*>(in + i);
            v[r][0] = t.x; v[r][1] = t.y; v[r][2] = t.z; v[r][3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[r][j] = i + j < n ? in[i + j] : 0u;
        }
        inc[r] = uint64_t(v[r][0]) + v[r][1] + v[r][2] + v[r][3];
    }
    uint64_t own[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) own[r] = inc[r];
    for (int off = 1; off < KMX_WAVE; off <<= 1) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint64_t t = __shfl_up(inc[r], off);
            if (lane >= uint32_t(off)) inc[r] += t;
        }
    }
    if (lane == KMX_WAVE - 1) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) wave_sum[r][w] = inc[r];
    }
    __syncthreads();
    uint64_t row_base;
    if constexpr (SPINE != 0) {
        __shared__ uint64_t part[KMX_BLOCK / KMX_WAVE];
        const uint64_t mine = SPINE == 2 ? uint64_t(blockIdx.x) * fine_per_tile : uint64_t(blockIdx.x);   // sums in front of this block
        uint64_t acc = 0;
        for (uint64_t i = threadIdx.x; i < mine; i += KMX_BLOCK) acc += bsum[i];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) part[w] = acc;
        __syncthreads();
        row_base = 0;
#pragma unroll
        for (uint32_t i = 0; i < KMX_BLOCK / KMX_WAVE; ++i) row_base += part[i];
        if (blockIdx.x + 1 == gridDim.x && threadIdx.x == 0) {
            uint64_t own = 0;
            if constexpr (SPINE == 2) { for (uint64_t i = mine; i < n_fine; ++i) own += bsum[i]; }
            else own = bsum[blockIdx.x];
            *total_out = row_base + own;
            if (pub.host) {
                // the batch's counters (k_lookup's, complete since the previous kernel, and the total just written) go to the
                // host from here, and the counter block of the NEXT batch on this handle starts from zero
#pragma unroll
                for (int c = 0; c < KMX_CTR_COUNT; ++c) {
                    pub.host[c] = c == KMX_CTR_TOTAL_HITS ? row_base + own : pub.cur[c];
                    pub.next[c] = 0;
                }
            }
        }
    } else {
        row_base = bsum[blockIdx.x];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        uint64_t carry = 0, total = 0;
#pragma unroll
        for (uint32_t i = 0; i < KMX_BLOCK / KMX_WAVE; ++i) {
            const uint64_t sw = wave_sum[r][i];
            if (i < w) carry += sw;
            total += sw;
        }
        uint64_t ex = row_base + carry + inc[r] - own[r];
        row_base += total;
        const uint64_t i0 = block_base + uint64_t(r) * (KMX_BLOCK * 4) + uint64_t(threadIdx.x) * 4;
        uint64_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t i = i0 + j;
            o[j] = ex;
            if constexpr (TILES) {
                if (i < n && v[r][j]) {
                    // boundaries x = t*tile with ex <= x < ex + v
                    uint64_t t = (ex + tile - 1) / tile;
                    const uint64_t t_end = (ex + v[r][j] - 1) / tile;
                    for (; t <= t_end && t <= n_tiles_cap; ++t) tile_q[t] = uint32_t(i);
                }
            }
            ex += v[r][j];
            if (i + 1 == n) {
                out[n] = ex;
                if constexpr (TILES) {
                    // boundaries at or beyond the total: the last query
                    for (uint64_t t = (ex + tile - 1) / tile; t <= n_tiles_cap && t <= (ex + tile - 1) / tile + 1; ++t) tile_q[t] = uint32_t(n - 1);
                }
            }
        }
        if (i0 + 3 < n) {
            typedef uint64_t u64x2s __attribute__((ext_vector_type(2)));
            u64x2s a, c;
            a[0] = o[0]; a[1] = o[1]; c[0] = o[2]; c[1] = o[3];
            *reinterpret_cast<u64x2s*>(out + i0) = a;
            *reinterpret_cast<u64x2s*>(out + i0 + 2) = c;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < n) out[i0 + j] = o[j];
        }
    }
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
}

// ---------------------------------------------------------------------------
// k_partition — for every tile boundary t*tile the query that owns that output
// slot: the last q with off[q] <= t*tile.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(KMX_BLOCK) void k_partition(const uint64_t* __restrict__ off, uint64_t nq,
                                                         uint64_t tile, uint64_t n_tiles,
                                                         uint32_t* __restrict__ tile_q)
{
    const uint64_t t = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    if (t > n_tiles) return;
    const uint64_t x = t * tile;
    uint64_t ub = upper_bound_dev<uint64_t>(off, nq + 1, x);   // first index with off > x
    uint64_t qi = ub ? ub - 1 : 0;
    if (qi > nq - 1) qi = nq - 1;
    tile_q[t] = uint32_t(qi);
}

// ---------------------------------------------------------------------------
// k_fill — materialises to_vector() (kmer_index_result.hpp:244-260) for EXACT
// and PREFIX queries, output-centric: one thread block owns KMX_FILL_TILE
// consecutive output slots whatever the bucket sizes are.
//   1. every non-empty query that starts inside the tile (plus the one that
//      covers its first slot) marks its first slot f in LDS and leaves
//      rec[f] = arena index of the element that belongs in slot f;
//   2. a max-scan of the marks tells every slot s which first slot f owns it;
//   3. slot s then copies arena[rec[f] + (s - f)] -> out[tile_base + s]: reads follow
//      the bucket runs (contiguous inside a run), writes are fully coalesced.
// ---------------------------------------------------------------------------

// rec_t = uint32_t when every arena index fits 31 bits (the usual case), uint64_t otherwise.
template <typename rec_t>
struct RecTraits {
    static constexpr rec_t SLOW = rec_t(1) << (sizeof(rec_t) * 8 - 1);
};

// One LDS word per output slot.  A query that owns slots of the tile writes, at its first slot f,
//     v = (arena index of the element that belongs in slot f) + (TILE - f)      (>= 1, top bit clear)
// or, for a query that is not a plain bucket copy, SLOW | (q - qa).  A "copy the last non-zero
// word" scan then hands every slot s its owner's word, and the element for s is
//     arena[v - TILE + s].
template <int E, bool NT, typename rec_t>
__global__ __launch_bounds__(KMX_BLOCK, (E <= 12 ? 8 : 6)) void k_fill(const KmxIndexDev* __restrict__ ix,
                                                    const uint32_t* __restrict__ arena,
                                                    const uint64_t* __restrict__ hit_off,
                                                    const uint32_t* __restrict__ tile_q,
                                                    const unsigned long long* __restrict__ total_dev,
                                                    QueryDesc d, uint32_t* __restrict__ out)
{
    constexpr int TILE = KMX_BLOCK * E;
    // the hit total lives in device memory so that the kernel can be launched before the host has read it;
    // the grid is then sized from an upper bound and surplus blocks leave here
    const uint64_t total = *total_dev;
    if (uint64_t(blockIdx.x) * TILE >= total) return;
    constexpr rec_t SLOW = RecTraits<rec_t>::SLOW;
    constexpr int VW = 16 / sizeof(rec_t);                     // words per 16-byte LDS access
    typedef rec_t recv_t __attribute__((ext_vector_type(VW)));
    __shared__ __attribute__((aligned(16))) rec_t word[TILE];
    __shared__ rec_t wave_tot[KMX_BLOCK / KMX_WAVE];
    __shared__ uint32_t tile_has_work;           // some slot of the tile is this kernel's to write

    const uint32_t tid = threadIdx.x;
    const uint64_t base = uint64_t(blockIdx.x) * TILE;
    const uint64_t tile_end = min(base + uint64_t(TILE), total);
    const uint32_t qa = tile_q[blockIdx.x], qb = tile_q[blockIdx.x + 1];

    // 1. clear (blocked, 16-byte LDS stores); then every query owning slots of this tile writes its word
    {
        recv_t* w4 = reinterpret_cast<recv_t*>(word) + tid * (E / VW);
#pragma unroll
        for (int j = 0; j < E / VW; ++j) w4[j] = recv_t(0);
        if (tid == 0) tile_has_work = 0;
    }
    __syncthreads();
    bool mine = false;
    for (uint64_t q = uint64_t(qa) + tid; q <= qb; q += KMX_BLOCK) {
        // three independent loads, issued together
        const uint64_t s = hit_off[q], e = hit_off[q + 1];
        const uint64_t sv = d.src[q];
        // non-short-circuit on purpose: sv takes part so that its load is issued with the other two
        if ((e > s) & (e > base) & (s < tile_end) & (sv != ~uint64_t(0))) {
            if (!(sv & SRC_SLOW)) {
                const uint32_t slot = s > base ? uint32_t(s - base) : 0u;
                word[slot] = rec_t(sv + (base + slot - s) + (TILE - slot));
                mine = true;
            } else if (sv & SRC_PREFIX) {
                // [s, s + len): the contiguous slice of every k-mer with this prefix -> plain copy;
                // [s + len, e): last-kmer positions (kmer_index.hpp:90-112) -> per-slot path
                // A slice of two or more runs is read, put in order and written by the prefix sort / merge kernels: its slots
                // here read the arena's padding, which says "do not store" (0xFFFFFFFF, KMX_ARENA_PAD).
                const uint64_t len = d.cnt[q] - uint32_t(__popcll(d.aux[q]));
                const uint64_t mid = s + len;
                if (len && mid > base) {
                    const uint32_t slot = s > base ? uint32_t(s - base) : 0u;
                    const bool sorted_elsewhere = len >= 2 && d.c0[q] >= 2;
                    word[slot] = sorted_elsewhere ? rec_t(ix->arena_elems + (TILE - slot))
                                                  : rec_t((sv & ~SRC_FLAGS) + (base + slot - s) + (TILE - slot));
                    mine |= !sorted_elsewhere;
                }
                if (e > mid && mid < tile_end) {
                    const uint32_t slot = mid > base ? uint32_t(mid - base) : 0u;
                    word[slot] = rec_t(SLOW | rec_t(q - qa));
                    mine = true;
                }
            } else {
                const uint32_t slot = s > base ? uint32_t(s - base) : 0u;
                word[slot] = rec_t(SLOW | rec_t(q - qa));              // STITCH: written by k_compact
                mine = true;
            }
        }
    }
    if (mine) tile_has_work = 1;
    __syncthreads();
    if (!tile_has_work) return;                  // a tile of nothing but slices the prefix kernels write (block-uniform)

    // 2. "last non-zero word so far" scan in blocked arrangement
    {
        rec_t v[E];
        recv_t* w4 = reinterpret_cast<recv_t*>(word) + tid * (E / VW);
#pragma unroll
        for (int j = 0; j < E / VW; ++j) {
            const recv_t t = w4[j];
#pragma unroll
            for (int i = 0; i < VW; ++i) v[VW * j + i] = t[i];
        }
        rec_t last = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) last = v[j] ? v[j] : last;
        // the nearest earlier lane that holds a word: one ballot + one cross-lane read
        const uint32_t lane = lane_id(), w = tid / KMX_WAVE;
        const uint64_t has = __ballot(last != 0);
        const uint64_t below = has & ((uint64_t(1) << lane) - 1);
        const int srcl = below ? 63 - __clzll(below) : 0;
        rec_t carry = __shfl(last, srcl);
        if (!below) carry = 0;
        if (lane == KMX_WAVE - 1) wave_tot[w] = last ? last : carry;
        __syncthreads();
        rec_t wcarry = 0;
#pragma unroll
        for (uint32_t i = 0; i < KMX_BLOCK / KMX_WAVE; ++i) {
            const rec_t t = wave_tot[i];
            if (i < w && t) wcarry = t;                         // ascending i: the nearest earlier wave wins
        }
        rec_t run = carry ? carry : wcarry;
#pragma unroll
        for (int j = 0; j < E; ++j) { run = v[j] ? v[j] : run; v[j] = run; }
#pragma unroll
        for (int j = 0; j < E / VW; ++j) {
            recv_t t;
#pragma unroll
            for (int i = 0; i < VW; ++i) t[i] = v[VW * j + i];
            w4[j] = t;
        }
    }
    __syncthreads();

    // rare path: PREFIX slots (dependent loads) and STITCH slots (the survivors of k_validate)
    auto slow_slot = [&](uint32_t slot, rec_t v, bool& live) -> uint32_t {
        const uint64_t q = uint64_t(qa) + uint64_t(v & ~SLOW);
        if (d.kind[q] != KMX_KIND_PREFIX) {
            // STITCH: the survivors k_validate left behind, or (no survivor buffer) nothing — k_compact writes the slot
            if (!d.stitch_hits) { live = false; return 0; }
            return d.stitch_hits[d.aux[q] * 64 + (base + slot - hit_off[q])];
        }
        // slice of every k-mer with this prefix, then the last-kmer offsets
        // (kmer_index.hpp:138-146): bit j of aux <-> position n - j
        const uint64_t idx = base + slot - hit_off[q];
        const uint64_t tmask = d.aux[q];
        const uint32_t len = d.cnt[q] - uint32_t(__popcll(tmask));
        if (idx < len) return arena[(d.src[q] & ~SRC_FLAGS) + idx];
        uint32_t t = uint32_t(idx - len);                               // t-th smallest position = t-th highest bit
        uint64_t mm = tmask;
        int bit = 63 - __clzll(mm);
        while (t--) { mm &= ~(uint64_t(1) << bit); bit = 63 - __clzll(mm); }
        return uint32_t(ix->n - uint64_t(bit));
    };

    // 3. gather, strided arrangement.  Branch-free straight-line code so that all E loads of the
    //    thread are in flight before the first store: dead or slow slots load arena[0] instead.
    //    Only the loaded value stays live per slot; the 32-bit variant addresses the arena with a
    //    32-bit byte offset against the uniform base pointer (arena < 4 GiB).
    uint32_t val[E];
    bool any_slow = false;
    if constexpr (sizeof(rec_t) == 4) {
        // buffer_load with a 32-bit voffset against a wave-uniform descriptor: one address VGPR per load
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t*>(arena), 0, int(uint32_t(ix->arena_elems * 4 + KMX_ARENA_PAD)), 0x00020000);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint32_t slot = j * KMX_BLOCK + tid;
            const bool live = base + slot < tile_end;
            const uint32_t wv = word[slot];
            const bool is_slow = (wv & SLOW) != 0;
            any_slow |= live && is_slow;
            const uint32_t boff = (live && !is_slow) ? ((wv - uint32_t(TILE) + slot) << 2) : 0u;
            val[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, boff, 0, 0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint32_t slot = j * KMX_BLOCK + tid;
            const bool live = base + slot < tile_end;
            const rec_t wv = word[slot];
            const bool is_slow = (wv & SLOW) != 0;
            any_slow |= live && is_slow;
            const uint64_t idx = (live && !is_slow) ? uint64_t(wv - rec_t(TILE) + slot) : 0;
            val[j] = arena[idx];
        }
    }
    if (__any(any_slow)) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint32_t slot = j * KMX_BLOCK + tid;
            const rec_t wv = word[slot];
            if (base + slot < tile_end && (wv & SLOW)) {
                bool live = true;
                const uint32_t x = slow_slot(slot, wv, live);
                val[j] = live ? x : 0xFFFFFFFFu;                        // 0xFFFFFFFF never is a position: "do not store"
            }
        }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const uint32_t slot = j * KMX_BLOCK + tid;
        if (base + slot < tile_end && val[j] != 0xFFFFFFFFu) {
            if constexpr (NT) __builtin_nontemporal_store(val[j], out + base + slot);
            else out[base + slot] = val[j];
        }
    }
}

// ---------------------------------------------------------------------------
// k_compact — STITCH queries: decode the mask words and compact the surviving
// candidates (is_valid + push_back of kmer_index_result.hpp:250-256) with a
// popcount prefix per word.  One wave per query; candidates are ascending, so
// the compacted list already is to_vector()'s sorted output.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(KMX_BLOCK) void k_compact(const uint32_t* __restrict__ arena, QueryDesc d,
                                                       uint64_t n_stitch,
                                                       const uint64_t* __restrict__ mask_words,
                                                       const uint64_t* __restrict__ hit_off,
                                                       uint32_t* __restrict__ out)
{
    // KMX_VGROUPS queries per wave, one per KMX_VGROUP-lane group (the chain list -> descriptor -> word ->
    // candidate -> store is latency bound, so several queries are kept in flight per wave)
    const uint32_t lane = lane_id();
    const uint32_t g = lane / KMX_VGROUP, gl = lane % KMX_VGROUP;
    const uint64_t wave = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    const uint64_t n_waves = uint64_t(gridDim.x) * (KMX_BLOCK / KMX_WAVE);
    for (uint64_t i0 = wave * KMX_VGROUPS; i0 < n_stitch; i0 += n_waves * KMX_VGROUPS) {
        const uint64_t i = i0 + g;
        const bool have = i < n_stitch;
        const uint32_t q = have ? d.stitch_list[i] : 0u;
        const uint32_t valid = have ? d.cnt[q] : 0u;
        const uint32_t c0 = (have && valid) ? d.c0[q] : 0u;               // nothing to copy when no candidate survived
        const uint64_t src = have ? (d.src[q] & ~SRC_FLAGS) : 0;
        const uint64_t* __restrict__ words = mask_words + (have ? d.aux[q] : 0);
        uint64_t o = have ? hit_off[q] : 0;
        const uint32_t n_it = (c0 + KMX_VGROUP - 1) / KMX_VGROUP;
        uint32_t max_it = n_it;
        for (int off = 32; off > 0; off >>= 1) max_it = max(max_it, uint32_t(__shfl_xor(int(max_it), off)));
        uint64_t word = 0;
        for (uint32_t it = 0; it < max_it; ++it) {
            if (it < n_it && (it % KMX_VSLICES) == 0) word = words[it / KMX_VSLICES];
            const uint32_t slice = uint32_t((word >> (KMX_VGROUP * (it % KMX_VSLICES))) & ((uint64_t(1) << KMX_VGROUP) - 1));
            if (it < n_it) {
                if ((slice >> gl) & 1u) {
                    const uint32_t rank = uint32_t(__popc(slice & ((1u << gl) - 1u)));
                    out[o + rank] = arena[src + uint64_t(it) * KMX_VGROUP + gl];
                }
                o += uint64_t(__popc(slice));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// merge_runs_lds — the std::sort of kmer_index_result.hpp:258 for a slice that is the concatenation of R ASCENDING runs
// (the buckets of consecutive keys, kmer_index.hpp:131-144): ceil(log2 R) rounds of pairwise merges in LDS, every round
// O(len) work — a merge, not a sort of unordered data.
//
// NT threads share one slice in `buf`.  A round merges the groups (2p, 2p+1) for every pair p: the pair's output is cut
// into chunks of E consecutive positions, one chunk per thread (sum over pairs of ceil(len_p / E) <= NT by the choice of
// E); a thread finds where its chunk starts in both groups (merge path: one binary search along the chunk's diagonal)
// and then merges E steps sequentially into registers; behind a barrier the registers go back to the buffer.  From the
// second round on the groups lie in a GAPPED layout — group g of the round at [bnd[g*w] + g, ...) with one cell holding
// 0xFFFFFFFF (never a position: n + k - 1 < 2^32, kmer_index.hpp:169-170) behind its last entry — so that a step needs
// no bounds: compare, min, advance one of two indices, one LDS read.  The first round reads the slice as it was staged
// (no gaps) and checks its indices instead.  An unpaired last group is merged with an empty one (= moved).
// buf: len + R + EMAX + 2 words at least; bnd: R + 1 run boundaries (bnd[0] = 0, bnd[R] = len); ptab: 4 * (pairs + 1)
// words.  R <= 2 * 64 pairs, and E = ceil(len / (NT - pairs)) <= EMAX is the caller's business.  On return (behind a
// sync) buf[0, len) is ascending.  tools/model_prefix_merge.py is a host model of the index arithmetic.
// ---------------------------------------------------------------------------
#define KMX_PM_SENT 0xFFFFFFFFu
// E sequential merge steps from the heads *pa / *pb into x[0, E).  A step: compare the heads, keep the smaller, advance that
// side, read its next entry — 8 VALU instructions and one LDS read when the groups end in sentinel cells; CHECKED (a slice as
// it was staged, no sentinels): the entry read at or behind a group's end counts as the sentinel (+3).
template <int EMAX, bool CHECKED>
__device__ __forceinline__ void merge_chunk(const KMX_LDS uint32_t* pa, const KMX_LDS uint32_t* pb, const KMX_LDS uint32_t* a_end,
                                            const KMX_LDS uint32_t* b_end, uint32_t E, uint32_t (&x)[EMAX])
{
    uint32_t va = *pa, vb = *pb;
    if (CHECKED) {
        va = pa < a_end ? va : KMX_PM_SENT;
        vb = pb < b_end ? vb : KMX_PM_SENT;
    }
#pragma unroll
    for (int j = 0; j < EMAX; ++j) {
        if (uint32_t(j) < E) {                                    // uniform over the cooperating threads
            const bool c = va < vb;
            x[j] = c ? va : vb;
            const KMX_LDS uint32_t* t = (c ? pa : pb) + 1;
            uint32_t nv = *t;
            if (CHECKED) nv = t < (c ? a_end : b_end) ? nv : KMX_PM_SENT;
            pa = c ? t : pa;
            pb = c ? pb : t;
            va = c ? nv : va;
            vb = c ? vb : nv;
        }
    }
}

// Where diagonal dg of the merge of A[0, na) and B[0, nb) crosses the merge path: the a in [max(0, dg - nb), min(dg, na)]
// with A[0, a) and B[0, dg - a) the dg smallest.  `steps` halving steps from a bracket of 2^steps - 1 (>= min(na, nb, dg) for
// every cooperating thread), branch-free: a candidate beyond the bracket reads a clamped entry and is refused.
__device__ __forceinline__ uint32_t merge_path_cut_lds(const KMX_LDS uint32_t* A, uint32_t na, const KMX_LDS uint32_t* B, uint32_t nb,
                                                       uint32_t dg, uint32_t steps)
{
    uint32_t lo = dg > nb ? dg - nb : 0u;
    const uint32_t hi = min(dg, na);
    const KMX_LDS uint32_t* Bd = B + dg;
    for (uint32_t st = steps ? 1u << (steps - 1) : 0u; st; st >>= 1) {
        const uint32_t cand = lo + st, cc = min(cand, hi);        // A[cand - 1] < B[dg - cand]: cand is not past the crossing
        const bool ok = (cand <= hi) & (A[int32_t(cc) - 1] < Bd[-int32_t(cc)]);
        lo = ok ? cand : lo;
    }
    return lo;
}

// E = ceil(len / (NT - ceil(R / 2))), rcp = floor(2^32 / E) + 1 (division by E as a multiplication: exact below 2^32 / E).
// The pair table of one round (w runs per group): lanes of ONE wave, lane p = pair p -> {start, middle, end, first chunk} at
// lpt[4 p ..], the round's chunk total at lpt[4 npairs + 3], the longest "shorter side" of any pair (bounds every merge-path
// bracket of the round) at lpt[4 npairs + 2].
__device__ __forceinline__ void merge_pair_table(KMX_LDS uint32_t* lpt, const KMX_LDS uint32_t* lbnd, uint32_t R, uint32_t w, uint32_t npairs,
                                                 uint32_t E, uint32_t rcp, uint32_t lane)
{
    uint32_t s = 0, mi = 0, e = 0, nch = 0;
    if (lane < npairs) {
        const uint32_t g0 = 2 * lane;
        s = lbnd[min(g0 * w, R)];
        mi = lbnd[min((g0 + 1) * w, R)];
        e = lbnd[min((g0 + 2) * w, R)];
        nch = __umulhi(e - s + E - 1, rcp);
    }
    uint32_t inc = nch, longest = min(mi - s, e - mi);
#pragma unroll
    for (uint32_t o = 1; o < KMX_WAVE; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
        longest = max(longest, uint32_t(__shfl_xor(longest, o)));
    }
    if (lane < npairs) {
        lpt[4 * lane] = s;
        lpt[4 * lane + 1] = mi;
        lpt[4 * lane + 2] = e;
        lpt[4 * lane + 3] = inc - nch;
    }
    if (lane == npairs - 1) lpt[4 * npairs + 3] = inc;
    if (lane == 0) lpt[4 * npairs + 2] = longest;
}

// C chunks merged by one thread in LOCKSTEP (C independent chains of dependent LDS reads: a merge step waits a full LDS round
// trip for the entry it has just chosen, so one chain per thread leaves the LDS pipe idle most of the time; with C chains the
// round trips overlap).  Same steps as merge_chunk.
template <int EMAX, bool CHECKED, int C>
__device__ __forceinline__ void merge_chunks(const KMX_LDS uint32_t* (&pa)[C], const KMX_LDS uint32_t* (&pb)[C], const KMX_LDS uint32_t* const (&a_end)[C],
                                             const KMX_LDS uint32_t* const (&b_end)[C], uint32_t E, uint32_t (&x)[C][EMAX])
{
    uint32_t va[C], vb[C];
#pragma unroll
    for (int u = 0; u < C; ++u) {
        va[u] = *pa[u]; vb[u] = *pb[u];
        if (CHECKED) {
            va[u] = pa[u] < a_end[u] ? va[u] : KMX_PM_SENT;
            vb[u] = pb[u] < b_end[u] ? vb[u] : KMX_PM_SENT;
        }
    }
    auto step = [&](int j) {
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const bool c = va[u] < vb[u];
            x[u][j] = c ? va[u] : vb[u];
            const KMX_LDS uint32_t* t = (c ? pa[u] : pb[u]) + 1;
            uint32_t nv = *t;
            if (CHECKED) nv = t < (c ? a_end[u] : b_end[u]) ? nv : KMX_PM_SENT;
            pa[u] = c ? t : pa[u];
            pb[u] = c ? pb[u] : t;
            va[u] = c ? nv : va[u];
            vb[u] = c ? vb[u] : nv;
        }
    };
    // E is odd and at least 3 (merge_runs_lds' callers): step 0, then two steps per (uniform) test of E
    static_assert(EMAX % 2 == 1 && EMAX >= 3, "an odd chunk length");
    step(0);
#pragma unroll
    for (int j = 1; j + 1 < EMAX; j += 2) {
        if (uint32_t(j + 1) < E) { step(j); step(j + 1); }
    }
}

// E = ceil(len / (C NT - ceil(R / 2))), rcp = floor(2^32 / E) + 1 (division by E as a multiplication: exact below 2^32 / E).
// TSTRIDE > 0 (blocks of several waves; ptab holds one table of TSTRIDE words per round, ceil(log2 R) of them): the pair
// tables of ALL rounds are written up front, wave r the table of round r, so that a round is partition - merge - barrier -
// write back - barrier with no serial table phase in it; TSTRIDE == 0 (one wave): the table of a round is made at its start.
// C: chunks per thread and round (thread t takes chunks t, t + NT, ...), merged in lockstep — see merge_chunks.
#ifdef KMX_PHASE_TIMING
// (measurement build only) thread 0 of a block adds the shader-clock cycles between two marks to a word of the index's debug block
#define KMX_MARK(word)                                                                                                \
    do {                                                                                                              \
        if (kmx_timing && threadIdx.x == 0) { const long long now__ = clock64(); atomicAdd(kmx_timing + (word), (unsigned long long)(now__ - kmx_t0)); kmx_t0 = now__; } \
    } while (0)
#else
#define KMX_MARK(word) do { } while (0)
#endif
// MergeRuns: the rounds one at a time — begin(), then round<true>() for the first and round<false>() for the others while more();
// merge_runs_lds below is that loop.  A caller that wants memory operations of its own to run under the rounds
// (k_prefix_merge_block) calls the rounds itself and passes two callables per round: b0 runs in front of the round's searches, b1 in
// front of its merge steps (every thread calls them; straight-line places).
struct MergeNoHook { __device__ __forceinline__ void operator()() const {} };
template <int EMAX, int NT, int TSTRIDE, int C>
struct MergeRuns {
    static constexpr bool PRETAB = TSTRIDE > 0;
    static constexpr uint32_t KMX_PM_TSTRIDE = TSTRIDE;
    KMX_LDS uint32_t* lb;
    const KMX_LDS uint32_t* lbnd;
    KMX_LDS uint32_t* lpt0;
    uint32_t R, E, rcp, tid, lane;
    uint32_t w = 1, ngroups, rnd = 0;
    unsigned long long* kmx_timing;
#ifdef KMX_PHASE_TIMING
    long long kmx_t0;
#endif
    __device__ __forceinline__ MergeRuns(uint32_t* buf, const uint32_t* bnd, uint32_t* ptab, uint32_t R_, uint32_t E_, uint32_t rcp_, uint32_t tid_,
                                         unsigned long long* timing)
        : lb((KMX_LDS uint32_t*)buf), lbnd((const KMX_LDS uint32_t*)bnd), lpt0((KMX_LDS uint32_t*)ptab), R(R_), E(E_), rcp(rcp_), tid(tid_),
          lane(tid_ & (KMX_WAVE - 1)), ngroups(R_), kmx_timing(timing)
    {
#ifdef KMX_PHASE_TIMING
        kmx_t0 = clock64();
#endif
    }
    __device__ __forceinline__ bool more() const { return ngroups > 1; }
    template <typename Sync>
    __device__ __forceinline__ void begin(Sync sync)
    {
        if (PRETAB) {
            uint32_t r = tid / KMX_WAVE, w2 = 1u << r;
            for (; w2 < R; r += NT / KMX_WAVE, w2 <<= NT / KMX_WAVE) {
                const uint32_t ng = (R + w2 - 1) / w2;
                merge_pair_table(lpt0 + r * KMX_PM_TSTRIDE, lbnd, R, w2, (ng + 1) >> 1, E, rcp, lane);
            }
            sync();
        }
        KMX_MARK(6);
    }
    template <bool FIRST, typename Sync, typename B0, typename B1>
    __device__ __forceinline__ void round(Sync sync, B0 b0, B1 b1)
    {
        constexpr bool first = FIRST;
        const uint32_t npairs = (ngroups + 1) >> 1;
        KMX_LDS uint32_t* lpt = PRETAB ? lpt0 + rnd * KMX_PM_TSTRIDE : lpt0;
        b0();
        if (!PRETAB) {
            if (tid < KMX_WAVE) merge_pair_table(lpt, lbnd, R, w, npairs, E, rcp, lane);
            sync();
            if (!first && tid < npairs && lpt[4 * tid + 2] == lpt[4 * tid + 1]) lb[lpt[4 * tid + 1] + 2 * tid + 1] = KMX_PM_SENT;   // the empty
            if (!first) sync();                                                                     // partner of an unpaired group
        }
        const uint32_t total = lpt[4 * npairs + 3];
        uint32_t steps = 0;
        {
            const uint32_t longest = __builtin_amdgcn_readfirstlane(lpt[4 * npairs + 2]);
            while ((1u << steps) <= longest) ++steps;
        }
        // this thread's C chunks of the round: chunk id -> its pair (a search over the pairs' first chunks), its diagonal
        uint32_t p[C], s[C], mi[C], e[C], dg[C], nout[C], na[C], nb[C];
#pragma unroll
        for (int u = 0; u < C; ++u) p[u] = 0;
        {
            uint32_t step = 1;
            while (step < npairs) step <<= 1;
            for (step >>= 1; step; step >>= 1) {
#pragma unroll
                for (int u = 0; u < C; ++u) {
                    const uint32_t cnd = p[u] + step;
                    if (cnd < npairs && lpt[4 * cnd + 3] <= tid + uint32_t(u) * NT) p[u] = cnd;
                }
            }
        }
        const KMX_LDS uint32_t* pA[C];
        const KMX_LDS uint32_t* pB[C];
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const bool active = tid + uint32_t(u) * NT < total;
            s[u] = lpt[4 * p[u]]; mi[u] = lpt[4 * p[u] + 1]; e[u] = lpt[4 * p[u] + 2];
            const uint32_t ch0 = lpt[4 * p[u] + 3];
            na[u] = active ? mi[u] - s[u] : 0u; nb[u] = active ? e[u] - mi[u] : 0u;
            dg[u] = active ? (tid + uint32_t(u) * NT - ch0) * E : 0u;         // the chunk's diagonal
            nout[u] = active ? min(E, na[u] + nb[u] - dg[u]) : 0u;
            pA[u] = lb + (first ? s[u] : s[u] + 2 * p[u]);
            pB[u] = lb + (first ? mi[u] : mi[u] + 2 * p[u] + 1);
        }
        // where every chunk's diagonal crosses the merge path: `steps` halving steps, the C searches in lockstep
        uint32_t lo[C];
        {
            // (every thread halves its OWN bracket: the probes of a wave's threads then fall wherever their diagonals put them.  With
            //  common strides — lo + 2^k for everybody — the probes of a step were a multiple of 2^k apart: one LDS bank for the
            //  whole wave in the steps of 32 words and more)
            uint32_t hi[C];
#pragma unroll
            for (int u = 0; u < C; ++u) { lo[u] = dg[u] > nb[u] ? dg[u] - nb[u] : 0u; hi[u] = min(dg[u], na[u]); }
            for (uint32_t it = 0; it < steps; ++it) {                 // steps: bits of the longest bracket of the round
                uint32_t av[C], bv[C], mid[C];
#pragma unroll
                for (int u = 0; u < C; ++u) {
                    mid[u] = (lo[u] + hi[u]) >> 1;                    // < hi while the bracket is open: A[mid], B[dg - 1 - mid] exist
                    av[u] = pA[u][mid[u]];
                    bv[u] = (pB[u] + dg[u])[-int32_t(mid[u]) - 1];
                }
#pragma unroll
                for (int u = 0; u < C; ++u) {
                    const bool open = lo[u] < hi[u], up = av[u] < bv[u];     // A[mid] < B[dg - 1 - mid]: the crossing is above mid
                    lo[u] = (open & up) ? mid[u] + 1 : lo[u];
                    hi[u] = (open & !up) ? mid[u] : hi[u];
                }
            }
        }
        KMX_MARK(7);                                               // pair lookup + merge-path searches
        b1();
        {
            uint32_t x[C][EMAX];
            const KMX_LDS uint32_t* qa[C];
            const KMX_LDS uint32_t* qb[C];
            const KMX_LDS uint32_t* ea[C];
            const KMX_LDS uint32_t* eb[C];
#pragma unroll
            for (int u = 0; u < C; ++u) { qa[u] = pA[u] + lo[u]; qb[u] = pB[u] + (dg[u] - lo[u]); ea[u] = lb + mi[u]; eb[u] = lb + e[u]; }
            merge_chunks<EMAX, FIRST, C>(qa, qb, ea, eb, E, x);    // (the first round's runs lie side by side: its reads are checked against their ends)
            KMX_MARK(8);                                           // the E merge steps
            sync();                                                // every read of the round is done
            KMX_MARK(9);                                           // barrier wait
#pragma unroll
            for (int u = 0; u < C; ++u) {
                KMX_LDS uint32_t* o = lb + (s[u] + p[u] + dg[u]);
                // (a chunk is whole — E outputs — unless it is the last of its pair: when every chunk of the wave is whole or
                //  idle the stores run under ONE predicate, not one per element)
                if (__all(nout[u] == E || nout[u] == 0)) {
                    if (nout[u]) {
                        o[0] = x[u][0];
#pragma unroll
                        for (int j = 1; j + 1 < EMAX; j += 2)              // (E is odd: pairs behind the first — one test, one two-word LDS store)
                            if (uint32_t(j + 1) < E) { o[j] = x[u][j]; o[j + 1] = x[u][j + 1]; }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < EMAX; ++j)
                        if (uint32_t(j) < E) {
                            if (uint32_t(j) < nout[u]) o[j] = x[u][j];
                        }
                }
            }
        }
        if (tid < npairs) lb[lpt[4 * tid + 2] + tid] = KMX_PM_SENT;   // the cell behind every merged group
        if (PRETAB && npairs > 1) {
            // the next round's unpaired group (if any) merges with an empty partner: its sentinel, in the layout just written
            const KMX_LDS uint32_t* nxt = lpt0 + (rnd + 1) * KMX_PM_TSTRIDE;
            const uint32_t np2 = (npairs + 1) >> 1;
            if (tid < np2 && nxt[4 * tid + 2] == nxt[4 * tid + 1]) lb[nxt[4 * tid + 1] + 2 * tid + 1] = KMX_PM_SENT;
        }
        sync();
        KMX_MARK(10);                                              // write back + sentinels + barrier
        w <<= 1;
        ngroups = npairs;
        ++rnd;
    }
};

template <int EMAX, int NT, int TSTRIDE, int C, typename Sync>
__device__ __forceinline__ void merge_runs_lds(uint32_t* buf, const uint32_t* bnd, uint32_t* ptab, uint32_t R, uint32_t len,
                                               uint32_t E, uint32_t rcp, uint32_t tid, Sync sync, unsigned long long* kmx_timing = nullptr)
{
    MergeRuns<EMAX, NT, TSTRIDE, C> m(buf, bnd, ptab, R, E, rcp, tid, kmx_timing);
    m.begin(sync);
    if (m.more()) m.template round<true>(sync, MergeNoHook(), MergeNoHook());
    while (m.more()) m.template round<false>(sync, MergeNoHook(), MergeNoHook());
}

// ---------------------------------------------------------------------------
// PREFIX queries whose slice is SEVERAL ascending runs (a slice that is one run — a prefix level's list, the only key of
// a range — is in order as k_fill copies it and appears on no work list).  By slice length:
//   <= 2048            one wave per query: k_prefix_sort_small (<= 4 runs / 512 positions from registers; many short runs
//                      by a bitonic network) or k_prefix_merge_small (merge_runs_lds)
//   <= 8192 / <= 32768 one 256- / 1024-thread block per query: k_prefix_sort_block (merge_runs_lds per chunk)
//   beyond             chunks of 32768 by k_prefix_sort_block, then k_prefix_merge_pass, ceil(log2 chunks) times
// All of them read the slice where it lies in the arena and write `out`; k_fill leaves those slots alone.
// k_prefix_len: tiles of the slices that need merge passes (0 for the others).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(KMX_BLOCK) void k_prefix_len(QueryDesc d, uint64_t n_prefix, const uint32_t* __restrict__ banded,
                                                          uint32_t* __restrict__ plen)
{
    const uint64_t i = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    if (i >= n_prefix) return;
    const uint32_t q = d.prefix_list[i];   // (the launcher passes the list region it wants walked)
    const uint32_t len = d.cnt[q] - uint32_t(__popcll(d.aux[q]));
    // large slices: output tiles of k_prefix_merge_pass (the scan of these is both the tile table and, times the tile, the
    // slice's place in the scratch buffer); the others are finished by k_prefix_sort_small / k_prefix_sort_block
    // (a slice of ONE run is in order as k_fill copies it: no chunks, no passes)
    // (... and a slice cut into bands is finished by k_prefix_merge_band: no chunks, no passes)
    plen[i] = (len > KMX_PSORT_BLOCK_CAP && d.c0[q] >= 2 && !banded[i]) ? (len + KMX_PM_TILE - 1) / KMX_PM_TILE : 0u;
}

// What the wave-per-slice kernels need to know about a listed PREFIX query before they can ask for its positions — six loads that
// depend on the list entry.  k_prefix_sort_small walks its share of the list with these a step ahead (and the list entry two): a slice of
// a few hundred positions used to start behind the whole chain of round trips (list entry -> descriptor -> planner -> offset table) — 7 % of
// the kernel on 381-position slices; k_prefix_merge_small, whose rounds are most of its time, gained nothing from the same.
struct PrefixHead {
    uint32_t R, len;
    uint64_t src, hoff, m, key;
};
__device__ __forceinline__ PrefixHead prefix_head(const QueryDesc& d, const uint64_t* __restrict__ qoff, const uint64_t* __restrict__ hit_off, uint32_t q)
{
    PrefixHead h;
    h.R = d.c0[q];
    h.len = d.cnt[q] - uint32_t(__popcll(d.aux[q]));
    h.src = d.src[q] & ~SRC_FLAGS;
    h.hoff = hit_off[q];
    h.m = qoff[q + 1] - qoff[q];
    h.key = d.key[q];
    return h;
}

// PREFIX queries with a short slice (<= KMX_PSORT_CAP positions), one wave per query.  The slice is the concatenation of
// R ascending runs (one per key of the prefix range) and is read where it lies in the arena (k_fill leaves the slots of
// a slice with two or more runs alone); what leaves for `out` is ascending — the std::sort of kmer_index_result.hpp:258.
// This kernel: up to four runs / 512 positions (what m = k - 1 yields on DNA) from registers, by rank in the sibling
// run; slices of many short runs by a bitonic network.  The class in between is k_prefix_merge_small's.
__global__ __launch_bounds__(KMX_BLOCK) void k_prefix_sort_small(const KmxIndexDev* __restrict__ ix,
                                                                 const uint64_t* __restrict__ qoff, QueryDesc d,
                                                                 uint64_t n_prefix,
                                                                 const uint64_t* __restrict__ hit_off,
                                                                 const uint32_t* __restrict__ arena,
                                                                 uint32_t* __restrict__ out)
{
    __shared__ uint32_t buf[KMX_BLOCK / KMX_WAVE][KMX_PSORT_CAP];
    __shared__ uint32_t bnd[KMX_BLOCK / KMX_WAVE][KMX_PSORT_MAX_RUNS + 1];
    const uint32_t lane = lane_id(), wv = threadIdx.x / KMX_WAVE;
    const uint64_t wave = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    const uint64_t n_waves = uint64_t(gridDim.x) * (KMX_BLOCK / KMX_WAVE);
    if (wave >= n_prefix) return;
    uint32_t q_ahead = d.prefix_list[wave];
    PrefixHead h_ahead = prefix_head(d, qoff, hit_off, q_ahead);
    q_ahead = wave + n_waves < n_prefix ? d.prefix_list[wave + n_waves] : 0u;
    for (uint64_t i = wave; i < n_prefix; i += n_waves) {
        const PrefixHead hd = h_ahead;                                  // this slice's (asked for a step ago)
        if (i + n_waves < n_prefix) h_ahead = prefix_head(d, qoff, hit_off, q_ahead);     // the next slice's, its list entry having arrived
        if (i + 2 * n_waves < n_prefix) q_ahead = d.prefix_list[i + 2 * n_waves];
        const uint32_t R = hd.R, len = hd.len;
        if (!KMX_PSORT_IS_SMALL(R, len) || KMX_PSORT_IS_MERGE(R, len) || R < 2 || len < 2) continue;   // wave-uniform
        const uint32_t* __restrict__ srcp = arena + hd.src;
        const KMX_GLOBAL uint32_t* offs = prefix_run_bounds(ix, hd.m, hd.key);
        if (lane <= min(R, uint32_t(KMX_PSORT_MAX_RUNS))) bnd[wv][lane] = offs[lane] - offs[0];
        uint32_t* __restrict__ seg = out + hd.hoff;
        auto wsync = [] {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        if (R > KMX_PSORT_MULTIWAY_RUNS) {
            // many short runs: a bitonic sort of the staged slice (independent compare-exchanges per stage)
            uint32_t n2 = 2;
            while (n2 < len) n2 <<= 1;
            for (uint32_t t = lane; t < n2; t += KMX_WAVE) buf[wv][t] = t < len ? srcp[t] : 0xFFFFFFFFu;
            wsync();
            bitonic_lds(buf[wv], n2, lane, uint32_t(KMX_WAVE), wsync);
            for (uint32_t t = lane; t < len; t += KMX_WAVE) seg[t] = buf[wv][t];
            wsync();
            continue;
        }
        wsync();                                                       // bnd[] is written
        if (len <= KMX_PSORT_PAIR_CAP) {
            // Up to four runs of fewer than 256 positions each (m = k - 1 on a text whose buckets hold about 100): every run
            // gets a slot of P entries in LDS (P = the power of two above the longest run), padded with 0xFFFFFFFF — the rank
            // of a position in its sibling run is then log2(P) branch-free halving steps from one pointer, no bounds to check
            // (4 VALU instructions per step and position instead of 8).  Round 1 merges runs (0,1) and (2,3) into two slots of
            // 2P entries, round 2 ranks every position in the other pair and writes it to its final place in global memory.
            const uint32_t e1 = bnd[wv][1], e2 = R >= 2 ? bnd[wv][2] : len, e3 = R >= 3 ? bnd[wv][3] : len, e4 = len;
            const uint32_t L0 = e1, L1 = e2 - e1, L2 = e3 - e2, L3 = e4 - e3;                     // wave-uniform
            const uint32_t longest = max(max(L0, L1), max(L2, L3));
            uint32_t P = 1;
            while (P <= longest) P <<= 1;
            if (8 * P <= KMX_PSORT_CAP) {
                constexpr int C = KMX_PSORT_PAIR_CAP / KMX_WAVE;
                uint32_t* __restrict__ b = buf[wv];
                uint32_t x[C], mi[C], gi[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const uint32_t t = lane + uint32_t(c) * KMX_WAVE;
                    gi[c] = uint32_t(t >= e1) + uint32_t(t >= e2) + uint32_t(t >= e3);              // run of t
                    const uint32_t gs = gi[c] == 0 ? 0u : gi[c] == 1 ? e1 : gi[c] == 2 ? e2 : e3;    // its start
                    mi[c] = t - gs;                                                                // index of t in its run
                    x[c] = srcp[min(t, len - 1)];
                }
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (lane + uint32_t(c) * KMX_WAVE < len) b[gi[c] * P + mi[c]] = x[c];
                for (uint32_t j = L0 + lane; j < P; j += KMX_WAVE) b[j] = 0xFFFFFFFFu;
                for (uint32_t j = L1 + lane; j < P; j += KMX_WAVE) b[P + j] = 0xFFFFFFFFu;
                for (uint32_t j = L2 + lane; j < P; j += KMX_WAVE) b[2 * P + j] = 0xFFFFFFFFu;
                for (uint32_t j = L3 + lane; j < P; j += KMX_WAVE) b[3 * P + j] = 0xFFFFFFFFu;
                wsync();
                {
                    const KMX_LDS uint32_t* at[C];
                    uint32_t tv[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) at[c] = (const KMX_LDS uint32_t*)b + (gi[c] ^ 1u) * P;   // the sibling run's slot
                    for (uint32_t st = P >> 1; st; st >>= 1) {
#pragma unroll
                        for (int c = 0; c < C; ++c) tv[c] = at[c][st - 1];
#pragma unroll
                        for (int c = 0; c < C; ++c) at[c] += tv[c] < x[c] ? st : 0u;
                    }
#pragma unroll
                    for (int c = 0; c < C; ++c) mi[c] += uint32_t(at[c] - ((const KMX_LDS uint32_t*)b + (gi[c] ^ 1u) * P));
                }
                if (R <= 2) {                                            // one pair: that was the whole sort
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        if (lane + uint32_t(c) * KMX_WAVE < len) seg[mi[c]] = x[c];
                    wsync();                                             // buf / bnd are reused by the next query
                    continue;
                }
                uint32_t* __restrict__ b2 = b + 4 * P;                   // the two merged pairs: slots of 2P entries
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (lane + uint32_t(c) * KMX_WAVE < len) b2[(gi[c] >> 1) * 2 * P + mi[c]] = x[c];
                for (uint32_t j = L0 + L1 + lane; j < 2 * P; j += KMX_WAVE) b2[j] = 0xFFFFFFFFu;
                for (uint32_t j = L2 + L3 + lane; j < 2 * P; j += KMX_WAVE) b2[2 * P + j] = 0xFFFFFFFFu;
                wsync();
                {
                    const KMX_LDS uint32_t* at[C];
                    uint32_t tv[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) at[c] = (const KMX_LDS uint32_t*)b2 + ((gi[c] >> 1) ^ 1u) * 2 * P;   // the other pair's slot
                    for (uint32_t st = P; st; st >>= 1) {
#pragma unroll
                        for (int c = 0; c < C; ++c) tv[c] = at[c][st - 1];
#pragma unroll
                        for (int c = 0; c < C; ++c) at[c] += tv[c] < x[c] ? st : 0u;
                    }
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const uint32_t rank = uint32_t(at[c] - ((const KMX_LDS uint32_t*)b2 + ((gi[c] >> 1) ^ 1u) * 2 * P));
                        // a position of pair 1 comes after every smaller one of pair 0 and after its own pair's smaller ones
                        if (lane + uint32_t(c) * KMX_WAVE < len) seg[mi[c] + rank] = x[c];
                    }
                }
                wsync();                                                 // buf / bnd are reused by the next query
                continue;
            }
        }
        for (uint32_t t = lane; t < len; t += KMX_WAVE) buf[wv][t] = srcp[t];
        wsync();
        if (len <= KMX_PSORT_PAIR_CAP) {
            // Up to four runs, every lane holds its (at most KMX_PSORT_PAIR_CAP / 64) positions in registers: the runs are
            // merged pairwise in place, two rounds at most.  A position's place in the merged pair is its index in its own
            // group plus its rank in the sibling group (no ties: positions are distinct); the searches of a lane's
            // positions advance in lockstep, a fixed number of branch-free halving steps.
            constexpr int C = KMX_PSORT_PAIR_CAP / KMX_WAVE;
            uint32_t* __restrict__ b = buf[wv];
            uint32_t e0 = bnd[wv][0], e1 = bnd[wv][1], e2 = R >= 2 ? bnd[wv][2] : len, e3 = R >= 3 ? bnd[wv][3] : len;
            const uint32_t e4 = len;
            for (uint32_t width = 1; width < R; width <<= 1) {
                uint32_t longest = max(max(e1 - e0, e2 - e1), max(e3 - e2, e4 - e3));     // wave-uniform
                uint32_t P = 1;
                while (P <= longest) P <<= 1;
                uint32_t x[C], own[C], s0[C], sn[C], cnt[C], tv[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const uint32_t t = lane + uint32_t(c) * KMX_WAVE;
                    const uint32_t gi = uint32_t(t >= e1) + uint32_t(t >= e2) + uint32_t(t >= e3);     // group of t
                    const uint32_t gs = gi == 0 ? e0 : gi == 1 ? e1 : gi == 2 ? e2 : e3;             // its start
                    const uint32_t ps = (gi & 2) ? e2 : e0;                                           // start of the merged pair
                    const uint32_t sb = (gi & 1) ? ps : (gi == 0 ? e1 : e3);                          // sibling: start ...
                    const uint32_t se = (gi & 1) ? gs : (gi == 0 ? e2 : e4);                          // ... and end
                    x[c] = b[min(t, len - 1)];
                    own[c] = ps + (t - gs);
                    s0[c] = sb;
                    sn[c] = se - sb;
                    cnt[c] = 0;
                }
                for (uint32_t st = P >> 1; st; st >>= 1) {
#pragma unroll
                    for (int c = 0; c < C; ++c) tv[c] = b[min(s0[c] + cnt[c] + st - 1, len - 1)];
#pragma unroll
                    for (int c = 0; c < C; ++c) cnt[c] += (cnt[c] + st - 1 < sn[c] && tv[c] < x[c]) ? st : 0u;
                }
                wsync();                                               // every read of this round is done
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (lane + uint32_t(c) * KMX_WAVE < len) b[own[c] + cnt[c]] = x[c];
                wsync();
                // the pairs are the groups of the next round
                e1 = e2; e2 = e4; e3 = e4;
            }
            for (uint32_t t = lane; t < len; t += KMX_WAVE) seg[t] = b[t];
            wsync();                                                   // buf/bnd are reused by the next query
            continue;
        }
        for (uint32_t t = lane; t < len; t += KMX_WAVE) {
            const uint32_t x = buf[wv][t];
            uint32_t pos = 0;
            for (uint32_t r = 0; r < R; ++r) {
                const uint32_t lo0 = bnd[wv][r], hi0 = bnd[wv][r + 1];
                if (t >= lo0 && t < hi0) { pos += t - lo0; continue; }   // own run: elements before me
                uint32_t lo = lo0, hi = hi0;                             // lower_bound of x in run r (no ties across runs)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (buf[wv][mid] < x) lo = mid + 1; else hi = mid;
                }
                pos += lo - lo0;
            }
            seg[pos] = x;
        }
        __builtin_amdgcn_wave_barrier();                                 // buf/bnd are reused by the next query
    }
}

// The small PREFIX slices in between: 2 .. KMX_PSORT_MAX_RUNS runs that average KMX_PMERGE_MIN_AVG positions or more (m = k - 2
// on DNA: 16 runs, 1526 positions at 1e8 letters) — merged, ceil(log2 R) rounds in LDS (merge_runs_lds), one wave per query.
__global__ __launch_bounds__(KMX_BLOCK) void k_prefix_merge_small(const KmxIndexDev* __restrict__ ix,
                                                                  const uint64_t* __restrict__ qoff, QueryDesc d,
                                                                  uint64_t n_prefix,
                                                                  const uint64_t* __restrict__ hit_off,
                                                                  const uint32_t* __restrict__ arena,
                                                                  uint32_t* __restrict__ out)
{
    __shared__ uint32_t buf[KMX_BLOCK / KMX_WAVE][KMX_PSORT_CAP + KMX_PM_WAVE_PAD];
    __shared__ uint32_t bnd[KMX_BLOCK / KMX_WAVE][KMX_PSORT_MAX_RUNS + 1];
    __shared__ uint32_t ptab[KMX_BLOCK / KMX_WAVE][4 * (KMX_PSORT_MAX_RUNS / 2 + 1)];
    const uint32_t lane = lane_id(), wv = threadIdx.x / KMX_WAVE;
    const uint64_t wave = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    const uint64_t n_waves = uint64_t(gridDim.x) * (KMX_BLOCK / KMX_WAVE);
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (uint64_t i = wave; i < n_prefix; i += n_waves) {
        // (everything about the query is the same in every lane: say so, the merge steps branch on it)
        const uint32_t q = __builtin_amdgcn_readfirstlane(d.prefix_list[i]);
        const uint32_t R = __builtin_amdgcn_readfirstlane(d.c0[q]);
        const uint32_t len = __builtin_amdgcn_readfirstlane(d.cnt[q] - uint32_t(__popcll(d.aux[q])));
        if (!KMX_PSORT_IS_MERGE(R, len)) continue;
        const uint32_t* __restrict__ srcp = arena + (d.src[q] & ~SRC_FLAGS);
        const KMX_GLOBAL uint32_t* offs = prefix_run_bounds(ix, qoff[q + 1] - qoff[q], d.key[q]);
        if (lane <= R) bnd[wv][lane] = offs[lane] - offs[0];
        for (uint32_t t = lane; t < len; t += KMX_WAVE) buf[wv][1 + t] = srcp[t];
        // (an ODD chunk length: the lanes' places in the buffer are E words apart — an even E maps them onto a fraction of the LDS banks)
        const uint32_t E = max(3u, ((len + (KMX_WAVE - (R + 1) / 2) - 1) / (KMX_WAVE - (R + 1) / 2)) | 1u);
        const uint32_t rcp = 0xFFFFFFFFu / E + 1;
        wsync();
        merge_runs_lds<KMX_PM_WAVE_EMAX, KMX_WAVE, 0, 1>(buf[wv] + 1, bnd[wv], ptab[wv], R, len, E, rcp, lane, wsync);
        uint32_t* __restrict__ seg = out + hit_off[q];
        for (uint32_t t = lane; t < len; t += KMX_WAVE) seg[t] = buf[wv][1 + t];
        wsync();                                                               // buf / bnd / ptab are the next query's
    }
}

// passes a slice of n_chunks sorted chunks needs until it is one ascending list
__device__ __forceinline__ uint32_t prefix_merge_passes(uint32_t len)
{
    const uint32_t n_chunks = (len + KMX_PSORT_BLOCK_CAP - 1) / KMX_PSORT_BLOCK_CAP;
    uint32_t p = 0;
    while ((1u << p) < n_chunks) ++p;
    return p;
}

// first index a in [max(0, dg - nb), min(dg, na)] with !(A[a] < B[dg - 1 - a]): where diagonal dg of the merge of A and B
// crosses the merge path (A[0, a) and B[0, dg - a) are the dg smallest)
template <typename PA, typename PB>
__device__ __forceinline__ uint32_t merge_path_cut(PA A, uint32_t na, PB B, uint32_t nb, uint32_t dg)
{
    uint32_t lo = dg > nb ? dg - nb : 0u, hi = min(dg, na);
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (A[mid] < B[dg - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The slice of a large PREFIX query is a row of KMX_PSORT_BLOCK_CAP-position chunks, each ascending after
// k_prefix_sort_block; pass t merges neighbouring groups of 2^t chunks — a merge of two ascending lists, O(len) per pass:
// a workgroup owns KMX_PM_TILE consecutive output positions (tiles never straddle a pair: pairs start at multiples of the
// chunk size), cuts both groups where the tile's two diagonals cross the merge path (two binary searches in global
// memory), stages the two pieces in LDS and merges them there, 16 outputs per thread, leaving coalesced.
// Buffers alternate between `out` (slice of query q at hit_off[q]) and `tmp` (at tile_off[i] * KMX_PM_TILE); a query with
// P passes starts in the buffer that makes its LAST pass land in `out` (k_prefix_sort_block writes there), so there is
// no copy back, and a query is not touched by passes beyond its own.
// merge_path_cut by a WAVE: 64 probes of the bracket per round trip instead of one (a tile of k_prefix_merge_pass started behind
// some thirty dependent round trips to global memory — its place in the tile table, its slice's descriptor, two merge-path
// searches of up to twenty steps by one thread each — and the passes ran at 2.6 TB/s algorithmic for it).  Returns the same cut in
// every lane.
template <typename PA, typename PB>
__device__ __forceinline__ uint32_t merge_path_cut_wave(PA A, uint32_t na, PB B, uint32_t nb, uint32_t dg, uint32_t lane)
{
    uint32_t lo = dg > nb ? dg - nb : 0u, hi = min(dg, na);      // the cut is in [lo, hi]; A[a] < B[dg - 1 - a] holds below it, not from it on
    while (lo < hi) {
        const uint32_t n = hi - lo, step = (n + KMX_WAVE - 1) / KMX_WAVE;
        const uint32_t a = lo + lane * step;                      // lane i asks about a_i = lo + i step (those below hi)
        const bool below = a < hi && A[a] < B[dg - 1 - a];
        const uint32_t c = uint32_t(__popcll(__ballot(below)));   // the a_i the cut lies above: a prefix of the lanes
        const uint32_t nlo = c ? lo + (c - 1) * step + 1 : lo;    // above a_(c-1) ...
        hi = min(hi, lo + c * step);                              // ... and not above a_c
        lo = nlo;
    }
    return lo;
}
// first index i in [0, n) with a[i] > x, by a wave (n >= 1, a ascending, a[0] <= x)
__device__ __forceinline__ uint64_t upper_bound_wave(const uint64_t* __restrict__ a, uint64_t n, uint64_t x, uint32_t lane)
{
    uint64_t lo = 0, hi = n;                                      // the answer is in [lo, hi]
    while (lo < hi) {
        const uint64_t m = hi - lo, step = (m + KMX_WAVE - 1) / KMX_WAVE;
        const uint64_t at = lo + lane * step;
        const bool le = at < hi && a[at] <= x;
        const uint64_t c = uint64_t(__popcll(__ballot(le)));
        const uint64_t nlo = c ? lo + (c - 1) * step + 1 : lo;
        hi = min(hi, lo + c * step);
        lo = nlo;
    }
    return lo;
}

__global__ __launch_bounds__(KMX_BLOCK) void k_prefix_merge_pass(QueryDesc d, uint64_t n_prefix,
                                                                 const uint64_t* __restrict__ tile_off,
                                                                 const uint64_t* __restrict__ hit_off,
                                                                 uint32_t* __restrict__ out, uint32_t* __restrict__ tmp,
                                                                 uint32_t pass)
{
    constexpr uint32_t T = KMX_PM_TILE, E = KMX_PM_TILE / KMX_BLOCK;
    __shared__ uint32_t buf[T + 2 + E];
    __shared__ uint32_t cut[2];
    const uint64_t b = blockIdx.x;
    if (b >= tile_off[n_prefix]) return;
    const uint64_t i = upper_bound_wave(tile_off, n_prefix + 1, b, lane_id()) - 1;        // (every wave of the block for itself: no barrier)
    const uint32_t q = d.prefix_list[i];
    const uint32_t len = d.cnt[q] - uint32_t(__popcll(d.aux[q]));
    const uint32_t P = prefix_merge_passes(len);
    if (pass >= P) return;
    const bool dst_is_out = ((P - pass) & 1u) != 0;
    uint32_t* __restrict__ s_out = out + hit_off[q];
    uint32_t* __restrict__ s_tmp = tmp + tile_off[i] * T;
    const uint32_t* __restrict__ src = dst_is_out ? s_tmp : s_out;
    uint32_t* __restrict__ dst = dst_is_out ? s_out : s_tmp;
    const uint32_t tid = threadIdx.x;
    const uint64_t G = uint64_t(KMX_PSORT_BLOCK_CAP) << pass;                   // positions per group
    const uint64_t o0 = (b - tile_off[i]) * T;                                  // first output of the tile, in the slice
    const uint64_t a_lo = (o0 / (2 * G)) * (2 * G);
    const uint64_t a_hi = min(a_lo + G, uint64_t(len)), b_hi = min(a_hi + G, uint64_t(len));
    const uint32_t na = uint32_t(a_hi - a_lo), nb = uint32_t(b_hi - a_hi);
    const uint32_t d0 = uint32_t(o0 - a_lo), d1 = min(d0 + T, na + nb);
    const uint32_t n_out = d1 - d0;
    if (nb == 0) {                                                               // no sibling group: carried over
        for (uint32_t t = tid; t < n_out; t += KMX_BLOCK) dst[o0 + t] = src[o0 + t];
        return;
    }
    const uint32_t* __restrict__ A = src + a_lo;
    const uint32_t* __restrict__ B = src + a_hi;
    if (tid < KMX_WAVE) {                                                        // wave 0: where the tile starts on the merge path ...
        const uint32_t c0 = merge_path_cut_wave(A, na, B, nb, d0, tid);
        if (tid == 0) cut[0] = c0;
    } else if (tid < 2 * KMX_WAVE) {                                             // ... wave 1: where it ends
        const uint32_t c1 = merge_path_cut_wave(A, na, B, nb, d1, tid - KMX_WAVE);
        if (tid == KMX_WAVE) cut[1] = c1;
    }
    __syncthreads();
    const uint32_t a0 = cut[0], a1 = cut[1], b0 = d0 - a0;
    const uint32_t nA = a1 - a0, nB = n_out - nA;
    // A's piece at [0, nA), a sentinel, B's piece at [nA + 1, nA + 1 + nB), a sentinel
    for (uint32_t t = tid; t < n_out + 2; t += KMX_BLOCK) {
        uint32_t v = KMX_PM_SENT;
        if (t < nA) v = A[a0 + t];
        else if (t > nA && t < n_out + 1) v = B[b0 + (t - nA - 1)];
        buf[t] = v;
    }
    __syncthreads();
    const KMX_LDS uint32_t* lb = (const KMX_LDS uint32_t*)buf;
    const uint32_t dg = min(tid * E, n_out);
    const uint32_t lo = merge_path_cut(lb, nA, lb + (nA + 1), nB, dg);
    uint32_t x[E];
    merge_chunk<int(E), false>(lb + lo, lb + (nA + 1 + dg - lo), nullptr, nullptr, E, x);
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < E; ++j)
        if (dg + j < n_out) buf[dg + j] = x[j];
    __syncthreads();
    for (uint32_t t = tid; t < n_out; t += KMX_BLOCK) dst[o0 + t] = buf[t];
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of the loaded code object: set it once per
// (kernel, device), not once per process (an index may live on any device of the node, replicas on several).
static void allow_big_lds(const void* fn, size_t bytes, int which)
{
    static std::atomic<uint64_t> done[3] = {{0}, {0}, {0}};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
    const uint64_t bit = uint64_t(1) << (dev & 63);
    if (done[which].load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));   // idempotent: a race sets it twice
    done[which].fetch_or(bit, std::memory_order_release);
}

static inline unsigned int blocks_for(uint64_t n, uint64_t per_block)
{
    uint64_t b = (n + per_block - 1) / per_block;
    return (unsigned int)(b ? b : 1);
}

uint64_t lookup_blocks(uint64_t nq, int items) { return blocks_for(nq, uint64_t(KMX_BLOCK) * uint64_t(items)); }

void launch_lookup(hipStream_t s, int items, bool pairs, const KmxIndexDev* ix, const uint8_t* qranks, const uint64_t* qoff,
                   uint64_t nq, const QueryDesc& d, unsigned long long* ctr, uint64_t* block_hits, uint32_t flags)
{
    const dim3 grid(blocks_for(nq, uint64_t(KMX_BLOCK) * uint64_t(items))), block(KMX_BLOCK);
    if (pairs) hipLaunchKernelGGL((k_lookup<4, true>), grid, block, 0, s, ix, qranks, qoff, nq, d, ctr, block_hits, flags);
    else if (items == 8) hipLaunchKernelGGL((k_lookup<8, false>), grid, block, 0, s, ix, qranks, qoff, nq, d, ctr, block_hits, flags);
    else hipLaunchKernelGGL((k_lookup<4, false>), grid, block, 0, s, ix, qranks, qoff, nq, d, ctr, block_hits, flags);
}

// The STITCH work list holds n_stitch queries from its front and n_tiny "tiny" ones from its back (list_end = one past
// the last entry of the buffer).  n_more = queries of the front list with further parts beyond the filter part: with a
// survivor buffer (d.stitch_hits) those are finished by k_validate_more from the survivor lists, without one k_validate
// checks them in line.
void launch_validate(hipStream_t s, const KmxIndexDev* ix, const uint32_t* arena, const uint8_t* qranks, const uint64_t* qoff,
                     const QueryDesc& d, uint64_t n_stitch, uint64_t n_more, uint64_t n_tiny, const uint32_t* tiny_list, uint64_t n_short,
                     uint64_t* mask_words, bool direct)
{
    if (n_tiny)
        hipLaunchKernelGGL(k_validate_tiny, dim3(blocks_for(n_tiny, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, ix, arena, qranks, qoff, d, tiny_list,
                           n_tiny, mask_words, direct && d.stitch_hits != nullptr);
    if (n_short) {
        const uint64_t swaves = (n_short + KMX_VGROUPS - 1) / KMX_VGROUPS;
        const unsigned int sblocks = (unsigned int)std::min<uint64_t>((swaves + 3) / 4, 256 * 32);
        hipLaunchKernelGGL(k_validate_short, dim3(sblocks ? sblocks : 1), dim3(KMX_BLOCK), 0, s, arena, d, n_short, mask_words);
    }
    if (!n_stitch) return;
    uint64_t waves = (n_stitch + KMX_VGROUPS - 1) / KMX_VGROUPS;        // KMX_VGROUPS queries per wave
    unsigned int blocks = (unsigned int)std::min<uint64_t>((waves + 3) / 4, 256 * 32);
    const dim3 grid(blocks ? blocks : 1), block(KMX_BLOCK);
    if (n_more && !d.stitch_hits) {
        hipLaunchKernelGGL(k_validate<true>, grid, block, 0, s, ix, arena, qranks, qoff, d, n_stitch, mask_words);
        return;
    }
    hipLaunchKernelGGL(k_validate<false>, grid, block, 0, s, ix, arena, qranks, qoff, d, n_stitch, mask_words);
    if (n_more) {
        // (n_more counts the queries with work behind k_validate<false>: further parts, big ones, wide filter buckets)
        const uint64_t sw = (n_stitch + KMX_VWIDE_SCAN - 1) / KMX_VWIDE_SCAN;
        const unsigned int sblocks = (unsigned int)std::min<uint64_t>((sw + 3) / 4, 256 * 16);
        hipLaunchKernelGGL(k_validate_wide, dim3(sblocks ? sblocks : 1), dim3(KMX_VWIDE_BLOCK), 0, s, arena, d, n_stitch, mask_words);
        hipLaunchKernelGGL(k_validate_more_thread, dim3(blocks_for(n_stitch, KMX_BLOCK)), block, 0, s, ix, arena, qranks, qoff, d, n_stitch,
                           uint32_t(n_more >= KMX_VMORE_THREAD_MANY ? KMX_VMORE_THREAD_PARTS_MANY : KMX_VMORE_THREAD_PARTS), mask_words);
        hipLaunchKernelGGL(k_validate_more, grid, block, 0, s, ix, arena, qranks, qoff, d, n_stitch, mask_words);
        const uint64_t wwaves = (n_stitch + KMX_VWAVE_SCAN - 1) / KMX_VWAVE_SCAN;
        const unsigned int wblocks = (unsigned int)std::min<uint64_t>((wwaves + 3) / 4, 256 * 16);
        hipLaunchKernelGGL(k_validate_wave, dim3(wblocks ? wblocks : 1), block, 0, s, ix, arena, qranks, qoff, d, n_stitch, mask_words);
    }
}

uint64_t scan_blocks(uint64_t n) { return blocks_for(n, KMX_SCAN_TILE); }

#define KMX_SCAN_FUSED_SPINE_BLOCKS 8192     // up to this many blocks the downsweep adds up the block sums itself

void launch_scan(hipStream_t s, const uint32_t* in, uint64_t n, uint64_t* bsum, uint64_t* out,
                 unsigned long long* total_out)
{
    const unsigned int nb = blocks_for(n, KMX_SCAN_TILE);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(KMX_BLOCK), 0, s, in, n, bsum);
    if (nb <= KMX_SCAN_FUSED_SPINE_BLOCKS) {
        hipLaunchKernelGGL((k_scan_down<false, 1>), dim3(nb), dim3(KMX_BLOCK), 0, s, in, n, bsum, out, uint64_t(1), uint64_t(0),
                           (uint32_t*)nullptr, total_out, uint64_t(0), CounterPub(), 0u);
        return;
    }
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(KMX_BLOCK), 0, s, bsum, uint64_t(nb), total_out);
    hipLaunchKernelGGL((k_scan_down<false, 0>), dim3(nb), dim3(KMX_BLOCK), 0, s, in, n, bsum, out, uint64_t(1), uint64_t(0),
                       (uint32_t*)nullptr, total_out, uint64_t(0), CounterPub(), 0u);
}

// scan + first-query-of-every-tile in one downsweep; tile_q must hold n_tiles_cap + 1 entries (nullptr: plain scan).
// lookup_sums: bsum already holds k_lookup's per-block sums of `in` (lookup_blocks(n) of them) — no reduce launch.
bool launch_scan_tiles(hipStream_t s, const uint32_t* in, uint64_t n, uint64_t* bsum, uint64_t* out,
                       unsigned long long* total_out, uint64_t tile, uint64_t n_tiles_cap, uint32_t* tile_q, int lookup_items,
                       const CounterPub& pub)
{
    const unsigned int nb = blocks_for(n, KMX_SCAN_TILE);
    const bool lookup_sums = lookup_items != 0;
    if (lookup_sums && nb <= KMX_SCAN_FUSED_SPINE_BLOCKS) {
        const uint64_t n_fine = lookup_blocks(n, lookup_items);
        const uint32_t fine = uint32_t(KMX_SCAN_TILE / (KMX_BLOCK * lookup_items));
        if (tile_q)
            hipLaunchKernelGGL((k_scan_down<true, 2>), dim3(nb), dim3(KMX_BLOCK), 0, s, in, n, bsum, out, tile, n_tiles_cap, tile_q, total_out, n_fine, pub, fine);
        else
            hipLaunchKernelGGL((k_scan_down<false, 2>), dim3(nb), dim3(KMX_BLOCK), 0, s, in, n, bsum, out, uint64_t(1), uint64_t(0),
                               (uint32_t*)nullptr, total_out, n_fine, pub, fine);
        return pub.host != nullptr;
    }
    if (!tile_q) { launch_scan(s, in, n, bsum, out, total_out); return false; }
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(KMX_BLOCK), 0, s, in, n, bsum);
    if (nb <= KMX_SCAN_FUSED_SPINE_BLOCKS) {
        hipLaunchKernelGGL((k_scan_down<true, 1>), dim3(nb), dim3(KMX_BLOCK), 0, s, in, n, bsum, out, tile, n_tiles_cap, tile_q, total_out, uint64_t(0), CounterPub(), 0u);
        return false;
    }
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(KMX_BLOCK), 0, s, bsum, uint64_t(nb), total_out);
    hipLaunchKernelGGL((k_scan_down<true, 0>), dim3(nb), dim3(KMX_BLOCK), 0, s, in, n, bsum, out, tile, n_tiles_cap, tile_q, total_out, uint64_t(0), CounterPub(), 0u);
    return false;
}

// fill variants, selected at run time (KMX_FILL_VARIANT, see kmx_capi.hip)
uint64_t fill_tile(const FillVariant& v) { return uint64_t(KMX_BLOCK) * v.e; }   // of an effective variant

void launch_partition(hipStream_t s, const uint64_t* off, uint64_t nq, uint64_t tile, uint64_t n_tiles, uint32_t* tile_q)
{
    hipLaunchKernelGGL(k_partition, dim3(blocks_for(n_tiles + 1, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, off, nq, tile, n_tiles, tile_q);
}

// The variant actually launched: 64-bit records (arenas >= 4 GiB) double the LDS per slot, so they always
// use 2048-slot tiles.
FillVariant effective_fill_variant(const FillVariant& v, bool rec32)
{
    FillVariant e = v;
    if (!rec32) e.e = 8;
    if (e.e != 4 && e.e != 8 && e.e != 12 && e.e != 16) e.e = 12;
    return e;
}

void launch_fill(hipStream_t s, const FillVariant& v0, bool rec32, const KmxIndexDev* ix, const uint32_t* arena, const uint64_t* hit_off,
                 const uint32_t* tile_q, const unsigned long long* total, uint64_t n_tiles, const QueryDesc& d, uint32_t* out)
{
    const FillVariant v = effective_fill_variant(v0, rec32);
    const dim3 grid((unsigned int)n_tiles), block(KMX_BLOCK);
    if (!rec32) {
        if (v.nt) hipLaunchKernelGGL((k_fill<8, true, uint64_t>), grid, block, 0, s, ix, arena, hit_off, tile_q, total, d, out);
        else hipLaunchKernelGGL((k_fill<8, false, uint64_t>), grid, block, 0, s, ix, arena, hit_off, tile_q, total, d, out);
        return;
    }
#define KMX_FILL_CASE(E_, NT_)                                                                                          \
    if (v.e == E_ && v.nt == NT_) {                                                                                     \
        hipLaunchKernelGGL((k_fill<E_, NT_, uint32_t>), grid, block, 0, s, ix, arena, hit_off, tile_q, total, d, out);   \
        return;                                                                                                         \
    }
    KMX_FILL_CASE(4, false)
    KMX_FILL_CASE(4, true)
    KMX_FILL_CASE(8, false)
    KMX_FILL_CASE(8, true)
    KMX_FILL_CASE(12, false)
    KMX_FILL_CASE(12, true)
    KMX_FILL_CASE(16, false)
    KMX_FILL_CASE(16, true)
#undef KMX_FILL_CASE
}

void launch_compact(hipStream_t s, const uint32_t* arena, const QueryDesc& d, uint64_t n_stitch,
                    const uint64_t* mask_words, const uint64_t* hit_off, uint32_t* out)
{
    const uint64_t cwaves = (n_stitch + KMX_VGROUPS - 1) / KMX_VGROUPS;
    unsigned int blocks = (unsigned int)std::min<uint64_t>((cwaves + 3) / 4, 256 * 32);
    hipLaunchKernelGGL(k_compact, dim3(blocks ? blocks : 1), dim3(KMX_BLOCK), 0, s, arena, d, n_stitch, mask_words, hit_off, out);
}

void launch_prefix_len(hipStream_t s, const QueryDesc& d, uint64_t n_prefix, const uint32_t* banded, uint32_t* plen)
{
    hipLaunchKernelGGL(k_prefix_len, dim3(blocks_for(n_prefix, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, d, n_prefix, banded, plen);
}


// PREFIX slices beyond k_prefix_sort_small's capacity: one block per chunk of up to KMX_PSORT_BLOCK_CAP positions (a slice
// up to that length is one chunk and done after this; longer ones: k_prefix_merge_pass).  The chunk is a row of
// ascending runs — the runs of the slice, cut at the chunk's ends — whose boundaries the block takes from the offset
// table the slice came from: up to RUNS of them are merged (merge_runs_lds), more are sorted by the bitonic network.
// Two shapes: 1024 threads around 128 KB of LDS (one block per CU), and, for slices up to KMX_PSORT_MID_CAP positions,
// 256 threads around 33 KB (four per CU: one block's staging and copy-out overlap the others' rounds).
#ifndef KMX_PSB_CPT
#define KMX_PSB_CPT 1          // chunks per thread and round in k_prefix_sort_block (merge_chunks: chains in lockstep; 2 measured slower:
#endif                         // the merge-path searches double and the LDS pipe, not its latency, is what the rounds wait for)
template <int THREADS, int CAP, int RUNS>
struct PsbShape {
    static constexpr int CPT = KMX_PSB_CPT;
    static constexpr int EMAX = ((CAP + (THREADS * CPT - RUNS / 2) - 1) / (THREADS * CPT - RUNS / 2)) | 1;
    static constexpr int WORDS = CAP + RUNS + EMAX + 5;      // staged chunk + sentinel cells + the reads of a chunk past its end
    static constexpr int TSTRIDE = 4 * (RUNS / 2 + 1);       // words of one round's pair table
    static constexpr int ROUNDS = RUNS <= 64 ? 6 : 7;        // ceil(log2 RUNS)
};
typedef PsbShape<1024, KMX_PSORT_BLOCK_CAP, 128> PsbBig;
#ifndef KMX_MID_THREADS
#define KMX_MID_THREADS 512    // threads of k_prefix_sort_block (slices up to KMX_PSORT_MID_CAP positions): eight waves per block and SIMD hide the
#endif                         // merge steps' LDS round trips better than four do (+9 % on 6 K slices; tools/exp/r04_mid_threads.sh)
#ifndef KMX_MID_OCC
#define KMX_MID_OCC (KMX_MID_THREADS > 256 ? 8 : 4)
#endif
#define KMX_BANDK_THREADS 256  // ... of k_prefix_merge_band (its gather needs the registers: 512 threads measured 3 % slower, 152 B of scratch)
typedef PsbShape<KMX_MID_THREADS, KMX_PSORT_MID_CAP, 64> PsbMid;
typedef PsbShape<KMX_BANDK_THREADS, KMX_PSORT_MID_CAP, 64> PsbBandK;

// distribute_sort_lds — the std::sort of kmer_index_result.hpp:258 for a chunk of MORE runs than merge_runs_lds takes (hundreds of
// short buckets: a sub-k query far below k on an element without prefix levels) whose positions are SPREAD over the text (the
// buckets of all k-mers with one prefix are: a k-mer family occurs all over a text without long repeats), as a distribution
// sort instead of a bitonic network over the whole chunk (measured: 256 runs of 95 positions 8.4 ms against 49 per 5e4 slices;
// for the few long runs the merge takes it was no faster, DESIGN A): a position's value bucket is (position * NB) / n — monotone —
// so (1) every thread counts its positions into the NB buckets (LDS atomics that return the slot inside the bucket), (2) a scan
// of the counts gives every bucket its place, (3) the positions are scattered to place + slot, (4) every bucket — three or four
// positions on average — is put in order from registers by an 8-input sorting network (insertion in LDS for the rare longer
// one), (5) the chunk leaves coalesced.  Returns false — nothing written but the counters — when some bucket holds more than
// KMX_PBK_GIVE_UP positions (a repeat of the text: the caller sorts the chunk with the bitonic network instead).
// out: CAP words, cnt: NB / 2 words (two 16-bit counters per word), wsum: THREADS / 64 + 2 words.
#define KMX_PBK_NB 8192
#define KMX_PBK_GIVE_UP 48
// The chunk arrives in REGISTERS: quad q of thread t holds positions 4 (q THREADS + t) .. + 3 (k_prefix_sort_items asks for a chunk's
// quads while the chunk before it is still being sorted).  after_scatter(): called by every thread once the quads have been read
// for the last time — the caller refills them there.
template <int THREADS, int CAP, typename After>
__device__ __forceinline__ bool distribute_sort_lds(uint32_t* __restrict__ out, uint32_t* __restrict__ cnt, uint32_t* __restrict__ wsum,
                                                    const u32x4 (&quads)[CAP / THREADS / 4], uint32_t c_len, uint32_t v_lo, uint64_t v_width, uint32_t tid,
                                                    After after_scatter)
{
    // (the positions lie in [v_lo, v_lo + v_width): the whole text for a chunk of a slice, a band's stretch of it for a band
    //  of k_prefix_split_*)
    constexpr int E = CAP / THREADS;                              // positions per thread
    constexpr int BPT = KMX_PBK_NB / THREADS;                     // buckets per thread in the scan and the bucket sorts
    static_assert(CAP % THREADS == 0 && KMX_PBK_NB % THREADS == 0 && BPT % 2 == 0 && BPT <= 8, "two counters per word, a thread's counters in one 16-byte read");
    static_assert(CAP <= 65535 && KMX_PBK_NB <= (1 << 13), "16-bit counters; bucket and slot share a word");
    const uint32_t mul = uint32_t(min((uint64_t(KMX_PBK_NB) << 32) / max(v_width, uint64_t(1)), uint64_t(0xFFFFFFFFu)));     // floor: ((p - lo) * mul) >> 32 < NB for every p - lo < width
    for (uint32_t i = tid; i < KMX_PBK_NB / 2; i += THREADS) cnt[i] = 0;
    if (tid == 0) wsum[THREADS / 64] = 0;                         // the longest bucket
    __syncthreads();
    // 1. count; the slot inside its bucket is all a position keeps (16 bits: two per register; its bucket is recomputed for the
    //    scatter)
    uint32_t slots[E / 2];
    static_assert(E % 4 == 0, "two slots per register, four positions per load");
#pragma unroll
    for (int q4 = 0; q4 < E / 4; ++q4) {
        const uint32_t i0 = uint32_t(q4) * 4u * THREADS + tid * 4u;
        const u32x4 v = quads[q4];
        const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            uint32_t packed = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t i = i0 + uint32_t(2 * h2 + h);
                uint32_t old = 0, sh = 0;
                if (i < c_len) {
                    const uint32_t b = min(__umulhi(pv[2 * h2 + h] - v_lo, mul), uint32_t(KMX_PBK_NB - 1));
                    sh = 16u * (b & 1u);
                    old = atomicAdd(&cnt[b >> 1], 1u << sh);
                }
                packed |= ((old >> sh) & 0xFFFFu) << (16 * h);
            }
            slots[2 * q4 + h2] = packed;
        }
    }
    __syncthreads();
    // 2. exclusive scan of the counts: thread t owns buckets [BPT t, BPT t + BPT)
    uint32_t c[BPT], st[BPT];
    {
        const uint32_t* mine = cnt + tid * (BPT / 2);
        uint32_t total = 0, longest = 0;
#pragma unroll
        for (int u = 0; u < BPT / 2; ++u) {
            const uint32_t w = mine[u];
            c[2 * u] = w & 0xFFFFu; c[2 * u + 1] = w >> 16;
        }
#pragma unroll
        for (int u = 0; u < BPT; ++u) { st[u] = total; total += c[u]; longest = max(longest, c[u]); }
        uint32_t inc = total;
        const uint32_t lane = tid & 63u, wv = tid / 64u;
#pragma unroll
        for (uint32_t o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
            longest = max(longest, uint32_t(__shfl_xor(int(longest), int(o))));
        }
        if (lane == 63) wsum[wv] = inc;
        if (lane == 0 && longest > KMX_PBK_GIVE_UP) atomicMax(&wsum[THREADS / 64], longest);
        __syncthreads();
        uint32_t carry = 0;
        for (uint32_t w2 = 0; w2 < wv; ++w2) carry += wsum[w2];
        const bool give_up = wsum[THREADS / 64] != 0;
        __syncthreads();                                          // (wsum is read: the counters may be overwritten by the places)
        if (give_up) return false;
        const uint32_t base = carry + inc - total;
#pragma unroll
        for (int u = 0; u < BPT; ++u) st[u] += base;
        uint32_t* mine_w = cnt + tid * (BPT / 2);
#pragma unroll
        for (int u = 0; u < BPT / 2; ++u) mine_w[u] = st[2 * u] | (st[2 * u + 1] << 16);      // places < CAP <= 65535
    }
    __syncthreads();
    // 3. scatter
#pragma unroll
    for (int q4 = 0; q4 < E / 4; ++q4) {
        const uint32_t i0 = uint32_t(q4) * 4u * THREADS + tid * 4u;
        const u32x4 v = quads[q4];
        const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int j = 4 * q4 + h;
            if (i0 + uint32_t(h) < c_len) {
                const uint32_t p = pv[h];
                const uint32_t b = min(__umulhi(p - v_lo, mul), uint32_t(KMX_PBK_NB - 1)), slot = (slots[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                const uint32_t place = (cnt[b >> 1] >> (16u * (b & 1u))) & 0xFFFFu;
                out[place + slot] = p;
            }
        }
    }
    after_scatter();
    __syncthreads();
    // 4. every bucket in order (a thread's BPT buckets are its own: no barrier between them)
#pragma unroll 1
    for (int u = 0; u < BPT; ++u) {
        const uint32_t n_b = c[u];
        uint32_t* __restrict__ bk = out + st[u];
        if (n_b >= 2 && n_b <= 8) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t t = 0; t < 8; ++t) v[t] = t < n_b ? bk[t] : 0xFFFFFFFFu;
            auto cx = [&](int a, int b2) { const uint32_t lo = min(v[a], v[b2]), hi = max(v[a], v[b2]); v[a] = lo; v[b2] = hi; };
            cx(0, 1); cx(2, 3); cx(4, 5); cx(6, 7);
            cx(0, 2); cx(1, 3); cx(4, 6); cx(5, 7);
            cx(1, 2); cx(5, 6); cx(0, 4); cx(3, 7);
            cx(1, 5); cx(2, 6);
            cx(1, 4); cx(3, 6);
            cx(2, 4); cx(3, 5);
            cx(3, 4);
#pragma unroll
            for (uint32_t t = 0; t < 8; ++t)
                if (t < n_b) bk[t] = v[t];
        } else if (n_b > 8) {
            for (uint32_t i = 1; i < n_b; ++i) {                 // (rare: insertion where it lies)
                const uint32_t key = bk[i];
                uint32_t j = i;
                while (j > 0 && bk[j - 1] > key) { bk[j] = bk[j - 1]; --j; }
                bk[j] = key;
            }
        }
    }
    __syncthreads();
    return true;
}

// One chunk of a slice through a block: `seg` the chunk where it lies in the arena, `dst` where it goes, runs[0 .. Rc] the Rc + 1
// entries of the offset table that bound its runs (entries are absolute: `base` = the table's value at the chunk's first position).
template <int THREADS, int CAP, int RUNS, bool MID>
__device__ __forceinline__ void psb_chunk(uint32_t* __restrict__ sbuf, uint32_t* __restrict__ bnd, uint32_t* __restrict__ ptab,
                                          const uint32_t* __restrict__ seg, uint32_t* __restrict__ dst, const KMX_GLOBAL uint32_t* runs,
                                          uint32_t base, uint32_t c_len, uint32_t Rc, uint32_t v_lo, uint64_t v_width, uint32_t tid)
{
    typedef PsbShape<THREADS, CAP, RUNS> Shape;
    const bool merge = Rc <= RUNS;
    static_assert(MID, "the 1024-thread shape has kernels of its own (k_prefix_merge_block, k_prefix_sort_items)");
    (void)v_lo; (void)v_width;
    if (merge && tid <= Rc) {
        const uint32_t o = runs[tid];
        bnd[tid] = o <= base ? 0u : min(o - base, c_len);
    }
    uint32_t n2 = 2;
    while (n2 < c_len) n2 <<= 1;
    const uint32_t n_stage = merge ? c_len : n2;
    constexpr uint32_t QPT = (CAP / 4 + THREADS - 1) / THREADS;       // quads per thread
    {
        // staging: four consecutive positions per thread and step as ONE 16-byte load (any alignment of the slice: global
        // loads only need 4 bytes) and one 16-byte LDS store, all of a thread's loads in flight before its first store —
        // a load-wait-store loop per position was 30 % of a 24 K slice's time (tools/probe_phases.py)
        u32x4 v[QPT];
#pragma unroll
        for (uint32_t u = 0; u < QPT; ++u) {
            const uint32_t t = (u * THREADS + tid) * 4;
            if (t + 3 < c_len) {
                v[u] = KMX_LOAD_QUAD(seg + t);
            } else {
                v[u].x = t + 0 < c_len ? seg[t + 0] : 0xFFFFFFFFu;
                v[u].y = t + 1 < c_len ? seg[t + 1] : 0xFFFFFFFFu;
                v[u].z = t + 2 < c_len ? seg[t + 2] : 0xFFFFFFFFu;
                v[u].w = 0xFFFFFFFFu;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < QPT; ++u) {
            const uint32_t t = (u * THREADS + tid) * 4;
            if (t < n_stage) *reinterpret_cast<u32x4*>(sbuf + t) = v[u];  // (up to three sentinels behind a merge's slice: inside its slack)
        }
    }
    __syncthreads();
    if (merge) {
        constexpr uint32_t CH = THREADS * Shape::CPT;                                     // chunks of a round
        const uint32_t E = max(3u, ((c_len + (CH - (Rc + 1) / 2) - 1) / (CH - (Rc + 1) / 2)) | 1u);   // (odd: LDS banks)
        merge_runs_lds<Shape::EMAX, THREADS, Shape::TSTRIDE, Shape::CPT>(sbuf, bnd, ptab, Rc, c_len, E, 0xFFFFFFFFu / E + 1, tid, [] { __syncthreads(); });
    } else {
        bitonic_lds(sbuf, n2, tid, uint32_t(THREADS), [] { __syncthreads(); });
    }
    {
        u32x4 v[QPT];
#pragma unroll
        for (uint32_t u = 0; u < QPT; ++u) {
            const uint32_t t = (u * THREADS + tid) * 4;
            if (t < c_len) v[u] = *reinterpret_cast<const u32x4*>(sbuf + t);
        }
#pragma unroll
        for (uint32_t u = 0; u < QPT; ++u) {
            const uint32_t t = (u * THREADS + tid) * 4;
            if (t + 3 < c_len) {
                KMX_STORE_QUAD(dst + t, v[u]);
            } else if (t < c_len) {
                dst[t] = v[u].x;
                if (t + 1 < c_len) dst[t + 1] = v[u].y;
                if (t + 2 < c_len) dst[t + 2] = v[u].z;
            }
        }
    }
    __syncthreads();
}

// The chunks of the longer slices (up to KMX_PSORT_BLOCK_CAP positions each) as ITEMS: what a 1024-thread block needs to know about
// one chunk, in one 32-byte record — written once by k_prefix_items (a wave per slice, a lane per chunk), so that the blocks
// that do the work find a chunk's header with ONE load asked for two chunks ahead instead of five dependent round trips (list
// entry -> descriptor -> planner -> offset table -> run search) in front of every chunk: 8 K of a 24 K-position slice's 90 K
// cycles (tools/probe_phases.py).  Chunks the merge takes (at most RUNS runs) are listed from the front of the item array,
// the others from its back.
struct PsbItem {
    uint64_t seg;                  // the chunk's first position in the arena
    uint64_t dst;                  // where it goes: an index into `out`, or with KMX_PSB_TMP into the scratch buffer of the merge passes
    const uint32_t* runs;          // offset-table entries of its runs (n_runs + 1 of them)
    uint32_t len_runs;             // length (16 bits; <= 32768) | min(runs, 0xFFFF) << 16
    uint32_t base;                 // the offset table's value at the chunk's first position
};
static_assert(sizeof(PsbItem) == 32, "one record = two 16-byte loads");
#define KMX_PSB_TMP (1ull << 63)

// The 256-thread shape: slices of at most KMX_PSORT_MID_CAP positions (one chunk each), a block per slice, four blocks per CU — from
// records as well (k_prefix_mid_items: a thread per listed slice, record i for list entry i, length 0 = not this shape's): a slice
// of six thousand positions took about as long to FIND (list entry -> descriptor -> planner -> offset table: five dependent round
// trips) as to merge; the record is one load, asked for a slice ahead.
__global__ __launch_bounds__(KMX_BLOCK) void k_prefix_mid_items(const KmxIndexDev* __restrict__ ix, const uint64_t* __restrict__ qoff, QueryDesc d,
                                                                uint64_t n_prefix, const uint64_t* __restrict__ hit_off, PsbItem* __restrict__ items)
{
    const uint64_t i = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    if (i >= n_prefix) return;
    const uint32_t q = d.prefix_list[i];
    const uint32_t R = d.c0[q];
    const uint32_t len = d.cnt[q] - uint32_t(__popcll(d.aux[q]));
    PsbItem it;
    it.seg = 0; it.dst = 0; it.runs = nullptr; it.len_runs = 0; it.base = 0;
    if (len <= KMX_PSORT_MID_CAP) {
        const KMX_GLOBAL uint32_t* offs = prefix_run_bounds(ix, qoff[q + 1] - qoff[q], d.key[q]);   // R + 1 run boundaries
        it.seg = d.src[q] & ~SRC_FLAGS;
        it.dst = hit_off[q];
        it.runs = (const uint32_t*)offs;
        it.len_runs = len | (min(R, 0xFFFFu) << 16);
        it.base = offs[0];
    }
    items[i] = it;
}

template <int THREADS, int CAP, int RUNS>
__global__ __launch_bounds__(THREADS, KMX_MID_OCC) void k_prefix_sort_block(const PsbItem* __restrict__ items, uint64_t n_prefix, uint64_t n_text,
                                                                const uint32_t* __restrict__ arena, uint32_t* __restrict__ out)
{
    typedef PsbShape<THREADS, CAP, RUNS> Shape;
    extern __shared__ __attribute__((aligned(16))) uint32_t sbuf[];             // Shape::WORDS
    __shared__ uint32_t bnd[RUNS + 1];
    __shared__ uint32_t ptab[Shape::ROUNDS * Shape::TSTRIDE];      // the pair tables of every round
    const uint32_t tid = threadIdx.x;
    uint64_t i = blockIdx.x;
    if (i >= n_prefix) return;
    PsbItem ahead = items[i];
    for (; i < n_prefix; i += gridDim.x) {
        const PsbItem it = ahead;
        if (i + gridDim.x < n_prefix) ahead = items[i + gridDim.x];                      // (the same in every lane; used a slice later)
        const uint32_t len = it.len_runs & 0xFFFFu;
        if (len == 0) continue;                                                          // another shape's (block-uniform)
        psb_chunk<THREADS, CAP, RUNS, true>(sbuf, bnd, ptab, arena + it.seg, out + it.dst, as_global(it.runs), it.base, len, it.len_runs >> 16, 0u, n_text, tid);
    }
}

template <int CAP, int RUNS>
__global__ __launch_bounds__(KMX_BLOCK) void k_prefix_items(const KmxIndexDev* __restrict__ ix, const uint64_t* __restrict__ qoff, QueryDesc d,
                                                            uint64_t n_prefix, const uint64_t* __restrict__ hit_off, const uint32_t* __restrict__ banded,
                                                            const uint64_t* __restrict__ tile_off, PsbItem* __restrict__ items, uint64_t cap_items,
                                                            unsigned long long* __restrict__ n_items)
{
    const uint32_t lane = lane_id();
    const uint64_t i = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    if (i >= n_prefix || banded[i]) return;                                               // (k_prefix_merge_band's)
    const uint32_t q = d.prefix_list[i];
    const uint32_t R = d.c0[q];
    const uint32_t len = d.cnt[q] - uint32_t(__popcll(d.aux[q]));
    if (len <= KMX_PSORT_MID_CAP) return;                                                 // the 256-thread kernel's
    const uint32_t n_chunks = (len + CAP - 1) / CAP;
    // where the chunks go: `out`, or the scratch buffer when the slice needs an odd number of merge passes
    const uint64_t dst0 = (prefix_merge_passes(len) & 1u) ? (tile_off[i] * KMX_PM_TILE) | KMX_PSB_TMP : hit_off[q];
    const uint64_t src0 = d.src[q] & ~SRC_FLAGS, n_text = ix->n;
    const KMX_GLOBAL uint32_t* offs = prefix_run_bounds(ix, qoff[q + 1] - qoff[q], d.key[q]);       // R + 1 run boundaries
    const uint32_t offs0 = offs[0];
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += KMX_WAVE) {
        const uint32_t c = c0 + lane;
        const bool active = c < n_chunks;
        const uint32_t c_lo = c * CAP, c_len = active ? min(uint32_t(CAP), len - c_lo) : 0u;
        uint32_t r0 = 0, Rc = R;
        if (n_chunks > 1 && active) {
            // the run that holds the chunk's first position, the first boundary at or behind its end
            r0 = uint32_t(upper_bound_dev<uint32_t>(offs, uint64_t(R) + 1, offs0 + c_lo)) - 1;
            Rc = uint32_t(lower_bound_dev<uint32_t>(offs, uint64_t(R) + 1, offs0 + c_lo + c_len)) - r0;
        }
        const bool merge = active && Rc <= RUNS;
        const uint64_t mb = __ballot(merge), ob = __ballot(active && !merge);
        unsigned long long bm = 0, bo = 0;
        if (lane == 0) {
            if (mb) bm = atomicAdd(&n_items[0], (unsigned long long)__popcll(mb));
            if (ob) bo = atomicAdd(&n_items[1], (unsigned long long)__popcll(ob));
        }
        bm = __shfl(bm, 0); bo = __shfl(bo, 0);
        if (active) {
            const uint64_t below = (uint64_t(1) << lane) - 1;
            const uint64_t at = merge ? bm + uint64_t(__popcll(mb & below)) : cap_items - 1 - (bo + uint64_t(__popcll(ob & below)));
            if (at < cap_items) {
                PsbItem it;
                it.seg = src0 + c_lo;
                it.dst = dst0 + c_lo;
                // (a chunk the merge does not take has no use for its run boundaries: the record says where its positions lie
                //  instead — anywhere in the text — as k_prefix_split_scan's records do for their bands)
                it.runs = merge ? (const uint32_t*)(offs + r0) : reinterpret_cast<const uint32_t*>(uintptr_t(0) | (uintptr_t(uint32_t(min(n_text, uint64_t(0xFFFFFFFFu)))) << 32));
                it.len_runs = c_len | (min(Rc, 0xFFFFu) << 16);
                it.base = offs0 + c_lo;
                items[at] = it;
            }
        }
    }
}

// The items the merge takes, one block per CU walking its share of them as a PIPELINE.  A 128-KB block owns its CU alone, so nothing
// else overlaps its memory phases, and a CU moves about 5 bytes per clock to or from memory (k_fill's rate as well): loading a
// 98-KB chunk, its header's five dependent round trips and storing it were half of the block's time with the LDS pipe idle
// (tools/probe_phases.py).  So a thread keeps two sets of eight quads in registers:
//   A   the chunk in hand: its positions as they arrived, staged into LDS; after the rounds its sorted positions, read back from LDS
//   B   while the rounds run: the PREVIOUS chunk's sorted positions, handed to memory quad by quad between the phases of the
//       rounds (merge_runs_lds' between()), each quad's registers asked to receive the NEXT chunk's positions right behind its store
// and the sets change places at the end; item records arrive two chunks ahead.  Nothing in the rounds waits for memory.
template <int THREADS, int CAP, int RUNS>
__global__ __launch_bounds__(THREADS, 4) void k_prefix_merge_block(const PsbItem* __restrict__ items, const unsigned long long* __restrict__ n_items_p,
                                                                 const uint32_t* __restrict__ arena, uint32_t* __restrict__ out,
                                                                 uint32_t* __restrict__ tmp, unsigned long long* kmx_timing)
{
    typedef PsbShape<THREADS, CAP, RUNS> Shape;
    extern __shared__ __attribute__((aligned(16))) uint32_t sbuf[];             // Shape::WORDS
    __shared__ uint32_t bnd[RUNS + 1];
    __shared__ uint32_t ptab[Shape::ROUNDS * Shape::TSTRIDE];      // the pair tables of every round
    constexpr uint32_t QPT = (CAP / 4 + THREADS - 1) / THREADS;       // quads per thread
    static_assert(QPT == 8, "between() below hands out eight quads");
    const uint32_t tid = threadIdx.x;
    const uint64_t n_items = *n_items_p, G = gridDim.x;
    uint64_t j = blockIdx.x;
    if (j >= n_items) return;
#ifdef KMX_PHASE_TIMING
    long long kmx_t0 = clock64();
#endif
    struct Hdr { const uint32_t* seg; uint32_t* dst; const KMX_GLOBAL uint32_t* runs; uint32_t c_len, Rc, base; };
    const u32x4* recs = reinterpret_cast<const u32x4*>(items);
    auto header = [&](const u32x4& a, const u32x4& b) {             // (a record is the same in every lane: say so)
        Hdr h;
        auto u = [](uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(x)); };            // (the builtin returns a SIGNED int)
        const uint64_t seg = uint64_t(u(a.x)) | uint64_t(u(a.y)) << 32;
        const uint64_t dst = uint64_t(u(a.z)) | uint64_t(u(a.w)) << 32;
        const uint64_t runs = uint64_t(u(b.x)) | uint64_t(u(b.y)) << 32;
        const uint32_t lr = u(b.z);
        h.seg = arena + seg;
        h.dst = (dst & KMX_PSB_TMP) ? tmp + (dst & ~KMX_PSB_TMP) : out + dst;
        h.runs = as_global(reinterpret_cast<const uint32_t*>(runs));
        h.c_len = lr & 0xFFFFu; h.Rc = lr >> 16;
        h.base = u(b.w);
        return h;
    };
    // (four times the thread's number, made opaque where it is used: every bound and address below derives from it, and the compiler
    //  would otherwise keep two dozen of them in registers across the rounds — the spills that follow are reloaded through the same
    //  counter the prefetched positions arrive on, i.e. they wait for the prefetch)
    auto tq4 = [&] { uint32_t x = tid * 4; asm volatile("" : "+v"(x)); return x; };
    // quad k of a chunk into registers.  Every thread inside the chunk asks for a whole quad, the one at the chunk's end too (the
    // arena is padded: up to three positions of whatever follows the chunk come along and are replaced by sentinels when the quad
    // is staged) — a branch for that thread with loads of its own would put a wait for ALL of the wave's loads into this place
    auto ask = [&](const Hdr& h, u32x4& q, uint32_t k) {
        const uint32_t t = tq4() + k * THREADS * 4;
        if (t < h.c_len) q = KMX_LOAD_QUAD(h.seg + t);
    };
    // quad k of a sorted chunk to where the chunk goes
    auto give = [&](const Hdr& h, const u32x4& q, uint32_t k) {
        const uint32_t t = tq4() + k * THREADS * 4;
        const int32_t l = int32_t(h.c_len) - int32_t(t);
        uint32_t* __restrict__ o = h.dst + t;
        if (l > 3) {
            KMX_STORE_QUAD(o, q);
        } else if (l > 0) {
            o[0] = q.x;
            if (l > 1) o[1] = q.y;
            if (l > 2) o[2] = q.z;
        }
    };
    u32x4 A[QPT], B[QPT];
    uint32_t bv = 0;
    Hdr cur = header(recs[2 * j], recs[2 * j + 1]), prev = cur;
    prev.c_len = 0;
    u32x4 na = {0, 0, 0, 0}, nb = {0, 0, 0, 0};                      // the record of item j + G, in flight
    if (j + G < n_items) { na = recs[2 * (j + G)]; nb = recs[2 * (j + G) + 1]; }
#pragma unroll
    for (uint32_t k = 0; k < QPT; ++k) ask(cur, A[k], k);
    if (tid <= cur.Rc) bv = cur.runs[tid];
    KMX_MARK(3);
    for (;;) {
        {
            const uint32_t tq = tq4();
            const int32_t left = int32_t(cur.c_len) - int32_t(tq);
            uint32_t* __restrict__ to = sbuf + tq;
#pragma unroll
            for (uint32_t u = 0; u < QPT; ++u) {
                const int32_t l = left - int32_t(u * THREADS * 4);
                if (l > 0) {
                    u32x4 w = A[u];
                    w.y = l > 1 ? w.y : 0xFFFFFFFFu;               // (up to three sentinels behind the chunk: inside the buffer's slack)
                    w.z = l > 2 ? w.z : 0xFFFFFFFFu;
                    w.w = l > 3 ? w.w : 0xFFFFFFFFu;
                    *reinterpret_cast<u32x4*>(to + u * THREADS * 4) = w;
                }
            }
        }
        if (tid <= cur.Rc) bnd[tid] = bv <= cur.base ? 0u : min(bv - cur.base, cur.c_len);
        __syncthreads();
        KMX_MARK(4);                                               // staging (the wait for the chunk's loads included)
        const bool more = j + G < n_items;                         // block-uniform
        Hdr nxt = cur;
        nxt.c_len = 0;
        if (more) {
            nxt = header(na, nb);
            if (tid <= nxt.Rc) bv = nxt.runs[tid];
            if (j + 2 * G < n_items) { na = recs[2 * (j + 2 * G)]; nb = recs[2 * (j + 2 * G) + 1]; }
        }
        // (no previous chunk: prev is a copy of cur with no positions; no next chunk: nxt is — so that the two steps are unconditional)
        auto quad = [&](uint32_t k) {
            give(prev, B[k], k);
            ask(nxt, B[k], k);
        };
        constexpr uint32_t CH = THREADS * Shape::CPT;                                     // chunks of a round
        const uint32_t E = max(3u, ((cur.c_len + (CH - (cur.Rc + 1) / 2) - 1) / (CH - (cur.Rc + 1) / 2)) | 1u);   // (odd: LDS banks)
        {
            // the first four rounds written out, each with two of B's quads at places of their own in the code (a quad's registers
            // must be NAMED where it is stored and asked for: picked by a counter they would be copied about, and a copy of a
            // register that a load is still on its way to waits for the load); rounds not run leave their quads for behind the loop
            auto sync = [] { __syncthreads(); };
            MergeRuns<Shape::EMAX, THREADS, Shape::TSTRIDE, Shape::CPT> m(sbuf, bnd, ptab, cur.Rc, E, 0xFFFFFFFFu / E + 1, tid, kmx_timing);
            m.begin(sync);
            uint32_t ran = 0;                                      // (uniform)
            if (m.more()) { m.template round<true>(sync, [&] { quad(0); }, [&] { quad(1); }); ran = 1; }
            if (m.more()) { m.template round<false>(sync, [&] { quad(2); }, [&] { quad(3); }); ran = 2; }
            if (m.more()) { m.template round<false>(sync, [&] { quad(4); }, [&] { quad(5); }); ran = 3; }
            if (m.more()) { m.template round<false>(sync, [&] { quad(6); }, [&] { quad(7); }); ran = 4; }
            while (m.more()) m.template round<false>(sync, MergeNoHook(), MergeNoHook());
#ifdef KMX_PHASE_TIMING
            kmx_t0 = clock64();
#endif
#pragma unroll
            for (uint32_t r = 0; r < QPT / 2; ++r)
                if (ran <= r) { quad(2 * r); quad(2 * r + 1); }
        }
        {
            const uint32_t tq = tq4();
            const int32_t left = int32_t(cur.c_len) - int32_t(tq);
            const uint32_t* __restrict__ from = sbuf + tq;
#pragma unroll
            for (uint32_t u = 0; u < QPT; ++u)
                if (left > int32_t(u * THREADS * 4)) A[u] = *reinterpret_cast<const u32x4*>(from + u * THREADS * 4);
        }
        __syncthreads();                                           // the buffer is the next chunk's
        KMX_MARK(5);                                               // sorted chunk LDS -> registers
        if (!more) {
#pragma unroll
            for (uint32_t k = 0; k < QPT; ++k) give(cur, A[k], k);
            break;
        }
#pragma unroll
        for (uint32_t k = 0; k < QPT; ++k) { const u32x4 t = A[k]; A[k] = B[k]; B[k] = t; }
        prev = cur;
        cur = nxt;
        j += G;
    }
}

// The items the merge does not take, from the back of the item array: chunks of more than RUNS runs (k_prefix_items), and the
// bands of the slices k_prefix_split_* spread by value (positions in no order at all, in the scratch buffer `split`) — distribution
// sort over the stretch of the text the record names, bitonic network where positions crowd.
template <int THREADS, int CAP, int RUNS>
__global__ __launch_bounds__(THREADS, 4) void k_prefix_sort_items(const PsbItem* __restrict__ items, uint64_t cap_items,
                                                                const unsigned long long* __restrict__ n_items_p,
                                                                const uint32_t* __restrict__ arena, const uint32_t* __restrict__ split,
                                                                uint32_t* __restrict__ out, uint32_t* __restrict__ tmp)
{
    typedef PsbShape<THREADS, CAP, RUNS> Shape;
    extern __shared__ __attribute__((aligned(16))) uint32_t sbuf[];             // Shape::WORDS + the distribution sort's counters
    __shared__ uint32_t wsum[THREADS / 64 + 2];
    constexpr uint32_t QPT = CAP / 4 / THREADS;                    // quads per thread
    const uint32_t tid = threadIdx.x;
    const uint64_t n_items = min((uint64_t)*n_items_p, cap_items), G = gridDim.x;
    uint64_t j = blockIdx.x;
    if (j >= n_items) return;
    struct Hdr { const uint32_t* seg; uint32_t* dst; uint32_t c_len, v_lo; uint64_t v_width; };
    auto header = [&](uint64_t jj) {
        const PsbItem it = items[cap_items - 1 - jj];              // (the same in every lane)
        Hdr h;
        h.dst = (it.dst & KMX_PSB_TMP) ? tmp + (it.dst & ~KMX_PSB_TMP) : out + it.dst;
        h.seg = (it.seg & KMX_PSB_TMP) ? split + (it.seg & ~KMX_PSB_TMP) : arena + it.seg;
        h.c_len = it.len_runs & 0xFFFFu;
        const uint64_t range = reinterpret_cast<uintptr_t>(it.runs);            // v_lo | v_width << 32 (k_prefix_items, k_prefix_split_scan)
        h.v_lo = uint32_t(range); h.v_width = range >> 32;
        return h;
    };
    // A 148-KB block owns its CU alone: the next chunk's positions are asked for (16-byte loads: the arena and the scratch buffer are
    // padded, the quad at a chunk's end is whole) as soon as this chunk's have been read for the last time, and arrive under its
    // bucket sorts and its copy-out.
    auto tq4 = [&] { uint32_t x = tid * 4; asm volatile("" : "+v"(x)); return x; };
    u32x4 v[QPT];
    auto ask = [&](const Hdr& h) {
        const uint32_t tq = tq4();
#pragma unroll
        for (uint32_t q = 0; q < QPT; ++q)
            if (tq + q * THREADS * 4 < h.c_len) v[q] = KMX_LOAD_QUAD(h.seg + tq + q * THREADS * 4);
    };
    Hdr cur = header(j);
    ask(cur);
    for (;;) {
        const bool more = j + G < n_items;                         // block-uniform
        Hdr nxt = cur;
        nxt.c_len = 0;
        if (more) nxt = header(j + G);
        const bool sorted = distribute_sort_lds<THREADS, CAP>(sbuf, sbuf + Shape::WORDS, wsum, v, cur.c_len, cur.v_lo, cur.v_width, tid, [&] { ask(nxt); });
        if (!sorted) {
            // positions that crowd (a repeat of the text): the bitonic network over the chunk, padded to a power of two
            uint32_t n2 = 2;
            while (n2 < cur.c_len) n2 <<= 1;
            const uint32_t tq = tq4();
#pragma unroll
            for (uint32_t q = 0; q < QPT; ++q) {
                const uint32_t t = tq + q * THREADS * 4;
                if (t < n2) {
                    u32x4 w = v[q];
                    w.x = t + 0 < cur.c_len ? w.x : 0xFFFFFFFFu;
                    w.y = t + 1 < cur.c_len ? w.y : 0xFFFFFFFFu;
                    w.z = t + 2 < cur.c_len ? w.z : 0xFFFFFFFFu;
                    w.w = t + 3 < cur.c_len ? w.w : 0xFFFFFFFFu;
                    *reinterpret_cast<u32x4*>(sbuf + t) = w;
                }
            }
            ask(nxt);
            __syncthreads();
            bitonic_lds(sbuf, n2, tid, uint32_t(THREADS), [] { __syncthreads(); });
        }
        {
            const uint32_t tq = tq4();
            const int32_t left = int32_t(cur.c_len) - int32_t(tq);
            const uint32_t* __restrict__ from = sbuf + tq;
            uint32_t* __restrict__ to = cur.dst + tq;
#pragma unroll
            for (uint32_t h0 = 0; h0 < QPT; h0 += QPT / 2) {
                u32x4 w[QPT / 2];
#pragma unroll
                for (uint32_t u = 0; u < QPT / 2; ++u)
                    if (left > int32_t((h0 + u) * THREADS * 4)) w[u] = *reinterpret_cast<const u32x4*>(from + (h0 + u) * THREADS * 4);
#pragma unroll
                for (uint32_t u = 0; u < QPT / 2; ++u) {
                    const int32_t l = left - int32_t((h0 + u) * THREADS * 4);
                    uint32_t* __restrict__ o = to + (h0 + u) * THREADS * 4;
                    if (l > 3) {
                        KMX_STORE_QUAD(o, w[u]);
                    } else if (l > 0) {
                        o[0] = w[u].x;
                        if (l > 1) o[1] = w[u].y;
                        if (l > 2) o[2] = w[u].z;
                    }
                }
            }
        }
        __syncthreads();
        if (!more) break;
        cur = nxt;
        j += G;
    }
}

// ---------------------------------------------------------------------------
// SPLITS.  A slice beyond one chunk whose runs are too many for bands (k_prefix_bands' cut table has a row per band of an entry per
// run) — a prefix far below k, or any prefix on an element of a large k, where every run is a position or two — went through
// chunks sorted in LDS and ceil(log2 chunks) merge passes, 8 bytes per position and pass (16 384 runs / 1.56 M positions: six
// passes, 8.2 of 10.3 ms).  Here its positions are spread by VALUE first, like the digits of a radix sort: band b of S takes the
// positions p with floor(p S / n) = b (about KMX_SPLIT of them), and every band is then ONE chunk for k_prefix_sort_items —
// 20 bytes per position in all, no passes.
//   k_prefix_bands        (the same wave that decides about bands) a record per slice, its counters, a list of its tiles
//   k_prefix_split_count  a block per tile of KMX_SPLIT_TILE positions: LDS histogram of the bands, added to the slice's counters
//   k_prefix_split_scan   a wave per slice: the bands' places (exclusive scan), one item per band — or, when a band would not fit a
//                         chunk (occurrences that cluster), the slice handed back to the chunks (banded[i] = 0)
//   k_prefix_split_scatter a block per tile again: positions binned by band in LDS, every bin appended to its band in `split`
// ---------------------------------------------------------------------------
#ifndef KMX_SPLIT
#define KMX_SPLIT 16384            // positions per band aimed at (a band takes up to KMX_PSORT_BLOCK_CAP; 6144 ... 28672 measured: tools/exp/r04_split_target.sh)
#endif
#define KMX_SPLIT_MAX 1024         // bands per slice at most (LDS histograms)
#ifndef KMX_SPLIT_TILE
#define KMX_SPLIT_TILE 8192        // positions per block of the count / scatter kernels
#endif
#ifndef KMX_SPLIT_SCATTER_THREADS
#define KMX_SPLIT_SCATTER_THREADS 1024
#endif
struct PsbSplit {
    uint64_t src0;                 // the slice's first position in the arena
    uint64_t dst0;                 // ... in `out`
    uint64_t tmp0;                 // ... in the scratch buffer
    uint32_t len, S;               // positions, bands (0: handed back to the chunks)
    uint32_t cnt_at;               // the slice's counters: S counts, S places, S fill cursors
    uint32_t list_i;               // its place in the work list (banded[list_i])
    uint32_t mul;                  // band of p = (p * mul) >> 32
    uint32_t pad;
};
static_assert(sizeof(PsbSplit) == 48, "three 16-byte loads");
struct PsbTile { uint32_t split, tile; };           // (split == 0xFFFFFFFF: no tile)
struct SplitRoom {                                  // what k_prefix_bands may hand out to the slices it sends this way (splits == nullptr: none)
    PsbSplit* splits; uint64_t cap_splits;
    PsbTile* tiles; uint64_t cap_tiles;
    uint64_t cap_counters, cap_scratch;             // words
};

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_prefix_split_count(const PsbSplit* __restrict__ splits, const PsbTile* __restrict__ tiles, uint64_t cap_tiles,
                                                                const unsigned long long* __restrict__ n_tiles_p, const uint32_t* __restrict__ arena,
                                                                uint32_t* __restrict__ counters)
{
    __shared__ uint32_t hist[KMX_SPLIT_MAX];
    const uint32_t tid = threadIdx.x;
    const uint64_t n_tiles = min((uint64_t)*n_tiles_p, cap_tiles);
    // (tiles that run at the same time should be tiles of DIFFERENT slices: every tile of a slice claims room on the slice's band
    //  cursors with atomics that return, and some two hundred tiles of one slice queueing on the same 64 addresses were what the
    //  scatter waited for — the list holds a slice's tiles one behind the other, so it is walked with a large odd stride)
    const uint64_t stride = n_tiles % 7919u ? 7919u : 7927u;
    for (uint64_t j0 = blockIdx.x; j0 < n_tiles; j0 += gridDim.x) {
        const uint64_t j = (j0 * stride) % n_tiles;
        const PsbTile t = tiles[j];
        if (t.split == 0xFFFFFFFFu) continue;                      // (block-uniform)
        const PsbSplit sp = splits[t.split];
        const uint32_t t0 = t.tile * KMX_SPLIT_TILE, n = min(uint32_t(KMX_SPLIT_TILE), sp.len - t0);
        for (uint32_t b = tid; b < sp.S; b += THREADS) hist[b] = 0;
        __syncthreads();
        const uint32_t* __restrict__ seg = arena + sp.src0 + t0;
        for (uint32_t e = tid * 4; e < n; e += THREADS * 4) {
            const u32x4 v = *reinterpret_cast<const u32x4_a4*>(seg + e);          // (the arena is padded: the quad at the tile's end is whole)
            atomicAdd(&hist[__umulhi(v.x, sp.mul)], 1u);
            if (e + 1 < n) atomicAdd(&hist[__umulhi(v.y, sp.mul)], 1u);
            if (e + 2 < n) atomicAdd(&hist[__umulhi(v.z, sp.mul)], 1u);
            if (e + 3 < n) atomicAdd(&hist[__umulhi(v.w, sp.mul)], 1u);
        }
        __syncthreads();
        for (uint32_t b = tid; b < sp.S; b += THREADS)
            if (hist[b]) atomicAdd(&counters[sp.cnt_at + b], hist[b]);
        __syncthreads();
    }
}

template <int CAP>
__global__ __launch_bounds__(KMX_BLOCK) void k_prefix_split_scan(PsbSplit* __restrict__ splits, uint64_t cap_splits, const unsigned long long* __restrict__ n_splits_p,
                                                                 uint32_t* __restrict__ counters, uint32_t* __restrict__ banded, uint64_t n_text,
                                                                 PsbItem* __restrict__ items, uint64_t cap_items, unsigned long long* __restrict__ n_other)
{
    const uint32_t lane = lane_id();
    const uint64_t w = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    if (w >= min((uint64_t)*n_splits_p, cap_splits)) return;
    PsbSplit sp = splits[w];
    if (sp.S == 0) return;                                         // (a record that was never written)
    uint32_t* __restrict__ cnt = counters + sp.cnt_at;
    uint32_t* __restrict__ place = cnt + sp.S;
    // the bands' places, and whether every band fits a chunk
    uint32_t carry = 0, biggest = 0;
    for (uint32_t b0 = 0; b0 < sp.S; b0 += KMX_WAVE) {
        const uint32_t b = b0 + lane, c = b < sp.S ? cnt[b] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (uint32_t o = 1; o < KMX_WAVE; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (b < sp.S) place[b] = carry + inc - c;
        carry += __shfl(inc, KMX_WAVE - 1);
        biggest = max(biggest, c);
    }
#pragma unroll
    for (uint32_t o = 1; o < KMX_WAVE; o <<= 1) biggest = max(biggest, uint32_t(__shfl_xor(int(biggest), int(o))));
    unsigned long long at = 0;
    const bool fits = biggest <= CAP && carry == sp.len;
    if (fits && lane == 0) at = atomicAdd(n_other, (unsigned long long)sp.S);
    at = __shfl(at, 0);
    if (!fits || at + sp.S > cap_items) {                          // occurrences cluster: chunks + merge passes for this slice
        if (lane == 0) { banded[sp.list_i] = 0; splits[w].S = 0; }
        return;
    }
    for (uint32_t b = lane; b < sp.S; b += KMX_WAVE) {
        // band b holds the positions p with (p * mul) >> 32 == b: from ceil(b 2^32 / mul) up to where band b + 1 starts
        const uint64_t lo = ((uint64_t(b) << 32) + sp.mul - 1) / sp.mul, hi = min(((uint64_t(b + 1) << 32) + sp.mul - 1) / sp.mul, n_text);
        PsbItem it;
        it.seg = (sp.tmp0 + place[b]) | KMX_PSB_TMP;
        it.dst = sp.dst0 + place[b];
        it.runs = reinterpret_cast<const uint32_t*>(uintptr_t(lo) | (uintptr_t(uint32_t(hi > lo ? hi - lo : 1)) << 32));
        it.len_runs = cnt[b] | (0xFFFFu << 16);
        it.base = 0;
        items[cap_items - 1 - (at + b)] = it;
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_prefix_split_scatter(const PsbSplit* __restrict__ splits, const PsbTile* __restrict__ tiles, uint64_t cap_tiles,
                                                                  const unsigned long long* __restrict__ n_tiles_p, const uint32_t* __restrict__ arena,
                                                                  uint32_t* __restrict__ counters, uint32_t* __restrict__ split)
{
    constexpr uint32_t PT = KMX_SPLIT_TILE / THREADS;              // positions per thread
    constexpr uint32_t BPT = KMX_SPLIT_MAX / THREADS;              // bands per thread in the scan
    __shared__ uint32_t hist[KMX_SPLIT_MAX];                       // positions of the tile per band, then the band's first slot in buf
    __shared__ uint32_t gat[KMX_SPLIT_MAX];                        // where the tile's share of a band goes in the scratch buffer, minus its first slot
    __shared__ uint32_t buf[KMX_SPLIT_TILE];
    __shared__ uint32_t wsum[THREADS / KMX_WAVE];
    const uint32_t tid = threadIdx.x, lane = tid & (KMX_WAVE - 1), wv = tid / KMX_WAVE;
    const uint64_t n_tiles = min((uint64_t)*n_tiles_p, cap_tiles);
    // (tiles that run at the same time should be tiles of DIFFERENT slices: every tile of a slice claims room on the slice's band
    //  cursors with atomics that return, and some two hundred tiles of one slice queueing on the same 64 addresses were what the
    //  scatter waited for — the list holds a slice's tiles one behind the other, so it is walked with a large odd stride)
    const uint64_t stride = n_tiles % 7919u ? 7919u : 7927u;
    for (uint64_t j0 = blockIdx.x; j0 < n_tiles; j0 += gridDim.x) {
        const uint64_t j = (j0 * stride) % n_tiles;
        const PsbTile t = tiles[j];
        if (t.split == 0xFFFFFFFFu) continue;                      // (block-uniform)
        const PsbSplit sp = splits[t.split];
        if (sp.S == 0) continue;                                   // handed back to the chunks (block-uniform)
        const uint32_t t0 = t.tile * KMX_SPLIT_TILE, n = min(uint32_t(KMX_SPLIT_TILE), sp.len - t0);
        for (uint32_t b = tid; b < KMX_SPLIT_MAX; b += THREADS) hist[b] = 0;
        __syncthreads();
        const uint32_t* __restrict__ seg = arena + sp.src0 + t0;
        // a position and its number inside its band's share of the tile.  Four consecutive positions a thread and step, as ONE 16-byte
        // load (the arena is padded: the quad at the tile's end is whole) — dword loads move at about 5 bytes per clock and CU, and
        // this kernel's loads, sorting and stores are phases of one block
        static_assert(PT % 4 == 0, "whole quads");
        uint32_t val[PT], slot[PT];
#pragma unroll
        for (uint32_t u = 0; u < PT; u += 4) {
            const uint32_t e = (u * THREADS) + tid * 4;
            if (e < n) {
                const u32x4 v = *reinterpret_cast<const u32x4_a4*>(seg + e);
                val[u] = v.x; val[u + 1] = v.y; val[u + 2] = v.z; val[u + 3] = v.w;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < PT; ++u) {
            const uint32_t e = (u / 4) * 4 * THREADS + tid * 4 + (u & 3u);
            if (e < n) slot[u] = atomicAdd(&hist[__umulhi(val[u], sp.mul)], 1u);
        }
        __syncthreads();
        {
            // exclusive scan of the counts (thread t: bands BPT t ..), and the tile's claim on every band it holds positions of
            uint32_t c[BPT], total = 0;
#pragma unroll
            for (uint32_t u = 0; u < BPT; ++u) { c[u] = hist[tid * BPT + u]; total += c[u]; }
            uint32_t inc = total;
#pragma unroll
            for (uint32_t o = 1; o < KMX_WAVE; o <<= 1) {
                const uint32_t x = __shfl_up(inc, o);
                if (lane >= o) inc += x;
            }
            if (lane == KMX_WAVE - 1) wsum[wv] = inc;
            __syncthreads();
            uint32_t first = inc - total;
            for (uint32_t w2 = 0; w2 < wv; ++w2) first += wsum[w2];
#pragma unroll
            for (uint32_t u = 0; u < BPT; ++u) {
                const uint32_t b = tid * BPT + u;
                hist[b] = first;
                if (c[u]) gat[b] = counters[sp.cnt_at + sp.S + b] + atomicAdd(&counters[sp.cnt_at + 2 * sp.S + b], c[u]) - first;
                first += c[u];
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < PT; ++u) {
            const uint32_t e = (u / 4) * 4 * THREADS + tid * 4 + (u & 3u);
            if (e < n) buf[hist[__umulhi(val[u], sp.mul)] + slot[u]] = val[u];
        }
        __syncthreads();
        uint32_t* __restrict__ to = split + sp.tmp0;
        for (uint32_t e = tid; e < n; e += THREADS) {              // (slot e of buf: consecutive slots of one band go to consecutive places)
            const uint32_t p = buf[e];
            to[gat[__umulhi(p, sp.mul)] + e] = p;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// BANDS.  A slice beyond the 256-thread shape whose runs are few (<= KMX_BAND_RUNS) is cut by VALUE, not by place: band s of S holds
// the positions in [s n / S, (s + 1) n / S) of every run — about len / S of them when the k-mers of the slice occur all over the
// text — so that the bands of a slice are independent merges whose outputs, one behind the other, ARE the sorted slice: no
// chunk of 32768 positions that owns a CU's LDS alone (its loads and stores overlap nothing: half the time of a 24 K slice),
// no merge passes behind the chunks of a longer slice.  A band is a 256-thread block's work (33 KB of LDS, four blocks per
// CU: one block's gather and copy-out run under the others' rounds).
//   k_prefix_bands       a wave per slice: where every run crosses every threshold (interpolation + galloping + halving search
//                        in the arena), the table of these cuts, one PsbBand per band — or, when some band would exceed
//                        KMX_PSORT_MID_CAP positions (a text whose occurrences cluster), nothing: the slice stays with the chunks
//   k_prefix_merge_band  a block per band: gathers the R run pieces into LDS, merge_runs_lds, leaves coalesced
// ---------------------------------------------------------------------------
#define KMX_BAND_FULL 7168         // positions per band tried first (7 / 8 of a block's capacity: 6.6 ms against 7.1 on 64 runs of 1526)
#ifndef KMX_BAND
#define KMX_BAND 6144              // positions per band aimed at (a band takes up to KMX_PSORT_MID_CAP: a third of slack for uneven texts)
#endif
#define KMX_BAND_RUNS 64           // the 256-thread shape's run capacity
#define KMX_BAND_MAX 512           // bands per slice at most (their sums live in LDS, a row per wave)
#ifndef KMX_PSORT_BAND_MIN
#define KMX_PSORT_BAND_MIN KMX_PSORT_BLOCK_CAP     // slices up to this length are not cut into bands
#endif
struct PsbBand {
    uint64_t src0;                 // the slice's first position in the arena
    uint64_t dst;                  // the band's first position in `out`
    const uint32_t* runs;          // offset-table entries of the slice's runs (n_runs + 1 of them)
    uint32_t cuts_at;              // row s of the slice's cut table (n_runs entries; row s + 1 follows it)
    uint32_t len_runs;             // positions in the band (16 bits) | runs << 16
};
static_assert(sizeof(PsbBand) == 32, "one record = two 16-byte loads");

// lower bound of `thr` in the ascending run A[0, n) of text positions (first index whose entry is not below thr): first probe
// where a text with evenly spread occurrences would have it, galloping from there, halving inside the bracket
__device__ __forceinline__ uint32_t band_cut(const KMX_GLOBAL uint32_t* A, uint32_t n, uint32_t thr, uint64_t n_text)
{
    if (n == 0) return 0;
    uint32_t lo = 0, hi = n;                                                              // the answer is in [lo, hi]
    uint32_t g = uint32_t(min(uint64_t(n - 1), (uint64_t(thr) * n) / n_text));
    uint32_t step = 1;
    if (A[g] < thr) {
        lo = g + 1;
        while (lo < hi) {
            const uint32_t p = min(lo + step - 1, hi - 1);
            if (A[p] < thr) { lo = p + 1; step <<= 1; } else { hi = p; break; }
        }
    } else {
        hi = g;
        while (lo < hi) {
            const uint32_t p = hi - min(step, hi - lo);
            if (A[p] < thr) { lo = p + 1; break; } else { hi = p; step <<= 1; }
        }
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (A[mid] < thr) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(KMX_BLOCK) void k_prefix_bands(const KmxIndexDev* __restrict__ ix, const uint64_t* __restrict__ qoff, QueryDesc d,
                                                            uint64_t n_prefix, const uint64_t* __restrict__ hit_off, const uint32_t* __restrict__ arena,
                                                            uint32_t* __restrict__ banded, PsbBand* __restrict__ bands, uint64_t cap_bands,
                                                            uint32_t* __restrict__ cuts, uint64_t cap_cuts, unsigned long long* __restrict__ used,
                                                            SplitRoom room)
{
    __shared__ uint32_t sums[KMX_BLOCK / KMX_WAVE][KMX_BAND_MAX + 1];                     // sums[s] = positions of the slice below threshold s
    const uint32_t lane = lane_id(), wv = threadIdx.x / KMX_WAVE;
    const uint64_t i = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    if (i >= n_prefix) return;                                                            // (wave-uniform, like everything about the slice)
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const uint32_t q = d.prefix_list[i];
    const uint32_t R = d.c0[q];
    const uint32_t len = d.cnt[q] - uint32_t(__popcll(d.aux[q]));
    const uint32_t S_room = (len + KMX_BAND - 1) / KMX_BAND;          // the most bands the slice may be cut into (the caller's room is for these)
    auto leave = [&] { if (lane == 0) banded[i] = 0; };
    // (a slice of one chunk stays a chunk: measured, 24 K positions of 16 runs take a 1024-thread block 5.5 ms per 5e4 slices and their
    //  four bands 5.2 + 1.2 for the cuts — the rounds, not the memory phases, are what both wait for; what the bands save is the
    //  merge passes behind the chunks of a longer slice: 98 K positions of 64 runs 10.1 -> 7.0 ms per 1e4 slices)
    if (len <= KMX_PSORT_BAND_MIN || R < 2 || R > KMX_BAND_RUNS || S_room > KMX_BAND_MAX) {
        // too many runs for bands, and more than one chunk: spread by value (k_prefix_split_*) — this wave makes the slice's record, its
        // counters' and its scratch space's places and the list of its tiles
        const uint32_t S2 = (len + KMX_SPLIT - 1) / KMX_SPLIT, n_tiles = (len + KMX_SPLIT_TILE - 1) / KMX_SPLIT_TILE;
        if (room.splits && len > KMX_PSORT_BLOCK_CAP && R > KMX_BAND_RUNS && S2 <= KMX_SPLIT_MAX) {
            unsigned long long si = 0, ca = 0, ta = 0, ti = 0;
            if (lane == 0) {
                si = atomicAdd(&used[2], 1ull);
                ca = atomicAdd(&used[3], 3ull * S2);
                ta = atomicAdd(&used[4], (unsigned long long)((len + 3u) & ~3u));
                ti = atomicAdd(&used[5], (unsigned long long)n_tiles);
            }
            si = __shfl(si, 0); ca = __shfl(ca, 0); ta = __shfl(ta, 0); ti = __shfl(ti, 0);
            // (the caller's room is an upper bound: these cannot fail; a record that is not written reads as 'no slice' — the caller
            //  clears both lists)
            if (si < room.cap_splits && ca + 3ull * S2 <= room.cap_counters && ta + len <= room.cap_scratch && ti + n_tiles <= room.cap_tiles) {
                if (lane == 0) {
                    PsbSplit sp;
                    sp.src0 = d.src[q] & ~SRC_FLAGS;
                    sp.dst0 = hit_off[q];
                    sp.tmp0 = ta;
                    sp.len = len; sp.S = S2;
                    sp.cnt_at = uint32_t(ca);
                    sp.list_i = uint32_t(i);
                    sp.mul = uint32_t((uint64_t(S2) << 32) / ix->n);                 // floor: (p * mul) >> 32 < S2 for every p < n
                    sp.pad = 0;
                    room.splits[si] = sp;
                    banded[i] = 2;
                }
                for (uint32_t t = lane; t < n_tiles; t += KMX_WAVE) room.tiles[ti + t] = PsbTile{uint32_t(si), t};
                return;
            }
        }
        leave();
        return;
    }
    // the slice's cut table: rows 0 .. S of R entries (row 0: zeros, row S: the runs' lengths)
    unsigned long long at = 0;
    if (lane == 0) at = atomicAdd(&used[1], (unsigned long long)(S_room + 1) * R);
    at = __shfl(at, 0);
    if (at + uint64_t(S_room + 1) * R > cap_cuts) { leave(); return; }
    uint32_t* __restrict__ tab = cuts + at;
    const KMX_GLOBAL uint32_t* offs = prefix_run_bounds(ix, qoff[q + 1] - qoff[q], d.key[q]);       // R + 1 run boundaries
    const uint32_t offs0 = offs[0];
    const uint64_t src0 = d.src[q] & ~SRC_FLAGS, n_text = ix->n;
    const KMX_GLOBAL uint32_t* seg = as_global(arena) + src0;
    // Fuller bands merge faster (a block's rounds cost about the same for 6 K positions as for 7 K): the slice is cut into bands of
    // KMX_BAND_FULL positions first and, when one of them would not fit a block — occurrences a little uneven over the text —, into
    // the KMX_BAND ones the room was made for; when those do not fit either the chunks take the slice.
    uint32_t S = 0;
    for (uint32_t target = KMX_BAND_FULL; ; target = KMX_BAND) {
        S = (len + target - 1) / target;
        for (uint32_t s = lane; s <= S; s += KMX_WAVE) sums[wv][s] = s == S ? len : 0u;
        for (uint32_t r = lane; r < R; r += KMX_WAVE) { tab[r] = 0; tab[S * R + r] = offs[r + 1] - offs[r]; }
        wsync();
        for (uint32_t p = lane; p < (S - 1) * R; p += KMX_WAVE) {
            const uint32_t s = 1 + p / R, r = p % R;
            const uint32_t r_at = offs[r] - offs0, r_len = offs[r + 1] - offs[r];
            const uint32_t c = band_cut(seg + r_at, r_len, uint32_t((uint64_t(s) * n_text) / S), n_text);
            tab[s * R + r] = c;
            atomicAdd(&sums[wv][s], c);
        }
        wsync();
        bool fits = true;
        for (uint32_t s = lane; s < S; s += KMX_WAVE) fits &= sums[wv][s + 1] - sums[wv][s] <= KMX_PSORT_MID_CAP;
        if (__all(fits)) break;
        if (target == KMX_BAND) { leave(); return; }                                      // occurrences cluster: the chunks take the slice
        wsync();                                                                          // (sums[] is rewritten)
    }
    unsigned long long b_at = 0;
    if (lane == 0) b_at = atomicAdd(&used[0], (unsigned long long)S);
    b_at = __shfl(b_at, 0);
    if (b_at + S > cap_bands) { leave(); return; }                                        // (cannot happen: the caller's bound is exact)
    const uint64_t dst0 = hit_off[q];
    for (uint32_t s = lane; s < S; s += KMX_WAVE) {
        PsbBand b;
        b.src0 = src0;
        b.dst = dst0 + sums[wv][s];
        b.runs = (const uint32_t*)offs;
        b.cuts_at = uint32_t(at + uint64_t(s) * R);
        b.len_runs = (sums[wv][s + 1] - sums[wv][s]) | (R << 16);
        bands[b_at + s] = b;
    }
    if (lane == 0) banded[i] = 1;
}

template <int THREADS, int CAP, int RUNS>
__global__ __launch_bounds__(THREADS, 4) void k_prefix_merge_band(const PsbBand* __restrict__ bands, uint64_t cap_bands,
                                                                const unsigned long long* __restrict__ n_bands_p, const uint32_t* __restrict__ cuts,
                                                                const uint32_t* __restrict__ arena, uint32_t* __restrict__ out)
{
    typedef PsbShape<THREADS, CAP, RUNS> Shape;
    extern __shared__ __attribute__((aligned(16))) uint32_t sbuf[];             // Shape::WORDS
    __shared__ uint32_t bnd[RUNS + 1];
    __shared__ uint32_t piece[RUNS];                               // where run r's piece of the band starts, in the slice
    __shared__ uint32_t ptab[Shape::ROUNDS * Shape::TSTRIDE];      // the pair tables of every round
    static_assert(RUNS <= KMX_WAVE, "one wave scans the pieces' lengths");
    const uint32_t tid = threadIdx.x;
    const uint64_t n_bands = min((uint64_t)*n_bands_p, cap_bands);
    for (uint64_t j = blockIdx.x; j < n_bands; j += gridDim.x) {
        const PsbBand b = bands[j];
        const uint32_t len = __builtin_amdgcn_readfirstlane(b.len_runs & 0xFFFFu), R = __builtin_amdgcn_readfirstlane(b.len_runs >> 16);
        if (len == 0) continue;                                    // (a stretch of the text without any occurrence: block-uniform)
        if (tid < KMX_WAVE) {
            uint32_t n_r = 0, at = 0;
            if (tid < R) {
                const KMX_GLOBAL uint32_t* runs = as_global(b.runs);
                const uint32_t lo = cuts[b.cuts_at + tid], hi = cuts[b.cuts_at + R + tid];
                n_r = hi - lo;
                at = (runs[tid] - runs[0]) + lo;
            }
            uint32_t inc = n_r;
#pragma unroll
            for (uint32_t o = 1; o < KMX_WAVE; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o);
                if (tid >= o) inc += t;
            }
            if (tid < R) { bnd[tid + 1] = inc; piece[tid] = at; }
            if (tid == 0) bnd[0] = 0;
        }
        __syncthreads();
        {
            // the gather: slot t of the band is entry t - bnd[r] of run r's piece (r advances with t: a thread's slots are THREADS apart).
            // (four consecutive slots a thread as one 16-byte load where they lie in one piece, position by position across a piece's
            //  end, was slower — 6.40 ms against 5.85 on 64 pieces of 95: nearly every wave has a lane at a piece's end every step)
            const uint32_t* __restrict__ seg = arena + b.src0;
            constexpr uint32_t PT = CAP / THREADS, GRP = 8;       // eight loads in flight per thread, then their eight LDS stores
            static_assert(PT % GRP == 0, "whole groups");
            uint32_t r = 0;
#pragma unroll 1
            for (uint32_t u0 = 0; u0 < PT && u0 * THREADS < len; u0 += GRP) {
                uint32_t val[GRP];
                const uint32_t t0 = u0 * THREADS + tid;
#pragma unroll
                for (uint32_t u = 0; u < GRP; ++u) {
                    const uint32_t t = t0 + u * THREADS;
                    if (t < len) {
                        while (t >= bnd[r + 1]) ++r;
                        val[u] = seg[piece[r] + (t - bnd[r])];
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < GRP; ++u) {
                    const uint32_t t = t0 + u * THREADS;
                    if (t < len) sbuf[t] = val[u];
                }
            }
        }
        __syncthreads();
        constexpr uint32_t CH = THREADS * Shape::CPT;                                     // chunks of a round
        const uint32_t E = max(3u, ((len + (CH - (R + 1) / 2) - 1) / (CH - (R + 1) / 2)) | 1u);   // (odd: LDS banks)
        merge_runs_lds<Shape::EMAX, THREADS, Shape::TSTRIDE, Shape::CPT>(sbuf, bnd, ptab, R, len, E, 0xFFFFFFFFu / E + 1, tid, [] { __syncthreads(); });
        {
            constexpr uint32_t QPT = (CAP / 4 + THREADS - 1) / THREADS;
            uint32_t* __restrict__ dst = out + b.dst;
            u32x4 v[QPT];
#pragma unroll
            for (uint32_t u = 0; u < QPT; ++u) {
                const uint32_t t = (u * THREADS + tid) * 4;
                if (t < len) v[u] = *reinterpret_cast<const u32x4*>(sbuf + t);
            }
#pragma unroll
            for (uint32_t u = 0; u < QPT; ++u) {
                const uint32_t t = (u * THREADS + tid) * 4;
                if (t + 3 < len) {
                    KMX_STORE_QUAD(dst + t, v[u]);
                } else if (t < len) {
                    dst[t] = v[u].x;
                    if (t + 1 < len) dst[t + 1] = v[u].y;
                    if (t + 2 < len) dst[t + 2] = v[u].z;
                }
            }
        }
        __syncthreads();
    }
}

uint64_t prefix_item_bytes() { return sizeof(PsbItem); }
uint64_t prefix_band_target() { return KMX_BAND; }
uint64_t prefix_band_min() { return KMX_PSORT_BAND_MIN; }
uint64_t prefix_band_runs() { return KMX_BAND_RUNS; }

// The slices beyond the 256-thread shape that can be cut into bands (few runs, occurrences spread over the text): cut tables and
// band records, banded[i] = 1 for them (0 for every other listed slice).  bands: room for cap_bands records of
// prefix_item_bytes() each, cap_bands >= 6 per listed slice beyond KMX_PSORT_MID_CAP + (positions of the slices beyond
// KMX_PSORT_BLOCK_CAP) / prefix_band_target(); cuts: cap_cuts words (a slice that finds no room stays with the chunks);
// used: two zeroed device counters (KMX_CTR_PSB_BANDS, KMX_CTR_PSB_CUTS)
void launch_prefix_bands(hipStream_t s, const KmxIndexDev* ix, const uint64_t* qoff, const QueryDesc& d, uint64_t n_prefix, const uint64_t* hit_off,
                         const uint32_t* arena, uint32_t* banded, void* bands, uint64_t cap_bands, uint32_t* cuts, uint64_t cap_cuts,
                         unsigned long long* used, const PrefixSplitRoom& sr)
{
    SplitRoom room{static_cast<PsbSplit*>(sr.splits), sr.cap_splits, static_cast<PsbTile*>(sr.tiles), sr.cap_tiles, sr.cap_counters, sr.cap_scratch};
    hipLaunchKernelGGL(k_prefix_bands, dim3(blocks_for(n_prefix * KMX_WAVE, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, ix, qoff, d, n_prefix, hit_off, arena,
                       banded, static_cast<PsbBand*>(bands), cap_bands, cuts, cap_cuts, used, room);
}

// the slices launch_prefix_bands sent to be spread by value: counted, placed (their bands become items at the back of `items`, or the
// slice goes back to the chunks: banded[i] = 0) and spread into sr.scratch.  used: the counters launch_prefix_bands counted into
void launch_prefix_split(hipStream_t s, const PrefixSplitRoom& sr, const unsigned long long* used, const uint32_t* arena, uint32_t* banded, uint64_t n_text,
                         void* items, uint64_t cap_items, unsigned long long* n_other)
{
    const unsigned int tb = (unsigned int)std::min<uint64_t>(std::max<uint64_t>(sr.cap_tiles, 1), 256 * 12);
    const PsbSplit* sp = static_cast<const PsbSplit*>(sr.splits);
    const PsbTile* tl = static_cast<const PsbTile*>(sr.tiles);
    hipLaunchKernelGGL((k_prefix_split_count<256>), dim3(tb), dim3(256), 0, s, sp, tl, sr.cap_tiles, used + 5, arena, sr.counters);
    hipLaunchKernelGGL((k_prefix_split_scan<KMX_PSORT_BLOCK_CAP>), dim3(blocks_for(sr.cap_splits * KMX_WAVE, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s,
                       static_cast<PsbSplit*>(sr.splits), sr.cap_splits, used + 2, sr.counters, banded, n_text, static_cast<PsbItem*>(items), cap_items, n_other);
    // (1024 threads per tile of 8192: eight positions a thread; measured against 256 / 512 threads and tiles of 2048 / 4096, tools/exp/r04_split_knobs.sh)
    hipLaunchKernelGGL((k_prefix_split_scatter<KMX_SPLIT_SCATTER_THREADS>), dim3(tb), dim3(KMX_SPLIT_SCATTER_THREADS), 0, s, sp, tl, sr.cap_tiles, used + 5, arena, sr.counters, sr.scratch);
}
uint64_t prefix_split_bytes(int what) { return what == 0 ? sizeof(PsbSplit) : sizeof(PsbTile); }
uint64_t prefix_split_target() { return KMX_SPLIT; }
uint64_t prefix_split_tile() { return KMX_SPLIT_TILE; }

// n_mid of the n_prefix listed queries have slices of at most KMX_PSORT_MID_CAP positions; banded / bands / cuts / n_bands: what
// launch_prefix_bands left; items: room for cap_items records (>= one per listed slice beyond KMX_PSORT_MID_CAP + one per
// KMX_PSORT_BLOCK_CAP positions of the slices longer than that); n_items: two zeroed counters
void launch_prefix_sort_block(hipStream_t s, const KmxIndexDev* ix, const uint64_t* qoff, const QueryDesc& d, uint64_t n_prefix, uint64_t n_mid,
                              const uint64_t* hit_off, const uint32_t* arena, uint32_t* out, const uint64_t* tile_off, uint32_t* tmp,
                              void* items, uint64_t cap_items, unsigned long long* n_items, const uint32_t* banded, const void* bands, uint64_t cap_bands,
                              const uint32_t* cuts, const unsigned long long* n_bands, const uint32_t* split, void* mid_items, uint64_t n_text,
                              unsigned long long* dbg)
{
    if (n_mid) {
        PsbItem* mi = static_cast<PsbItem*>(mid_items);
        hipLaunchKernelGGL(k_prefix_mid_items, dim3(blocks_for(n_prefix, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, ix, qoff, d, n_prefix, hit_off, mi);
        auto fn = k_prefix_sort_block<KMX_MID_THREADS, KMX_PSORT_MID_CAP, 64>;
        const size_t lds = size_t(PsbMid::WORDS) * 4;
        const unsigned int blocks = (unsigned int)std::min<uint64_t>(n_prefix, 256 * 16);
        hipLaunchKernelGGL(fn, dim3(blocks ? blocks : 1, 1), dim3(KMX_MID_THREADS), lds, s, mi, n_prefix, n_text, arena, out);
    }
    if (n_prefix > n_mid) {
        {
            auto fn = k_prefix_merge_band<KMX_BANDK_THREADS, KMX_PSORT_MID_CAP, KMX_BAND_RUNS>;
            const size_t lds = size_t(PsbBandK::WORDS) * 4;
            const unsigned int blocks = (unsigned int)std::min<uint64_t>(cap_bands, 256 * 16);
            hipLaunchKernelGGL(fn, dim3(blocks ? blocks : 1), dim3(KMX_BANDK_THREADS), lds, s, static_cast<const PsbBand*>(bands), cap_bands, n_bands, cuts, arena, out);
        }
        PsbItem* it = static_cast<PsbItem*>(items);
        hipLaunchKernelGGL((k_prefix_items<KMX_PSORT_BLOCK_CAP, 128>), dim3(blocks_for(n_prefix * KMX_WAVE, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, ix, qoff, d,
                           n_prefix, hit_off, banded, tile_off, it, cap_items, n_items);
        const unsigned int blocks = (unsigned int)std::min<uint64_t>(cap_items, 256);     // one per CU: they are pipelines, not a queue
        {
            auto fn = k_prefix_merge_block<1024, KMX_PSORT_BLOCK_CAP, 128>;
            const size_t lds = size_t(PsbBig::WORDS) * 4;
            allow_big_lds(reinterpret_cast<const void*>(fn), lds, 0);
            hipLaunchKernelGGL(fn, dim3(blocks ? blocks : 1), dim3(1024), lds, s, it, n_items, arena, out, tmp, dbg);
        }
        {
            auto fn = k_prefix_sort_items<1024, KMX_PSORT_BLOCK_CAP, 128>;
            const size_t lds = size_t(PsbBig::WORDS + KMX_PBK_NB / 2) * 4;        // (+ the counters of the distribution sort)
            allow_big_lds(reinterpret_cast<const void*>(fn), lds, 2);
            hipLaunchKernelGGL(fn, dim3(blocks ? blocks : 1), dim3(1024), lds, s, it, cap_items, n_items + 1, arena, split, out, tmp);
        }
    }
}

// ---------------------------------------------------------------------------
// Index construction on the device — the work of kmer_index_element::create
// (kmer_index.hpp:154-179) for key spaces that fit a histogram:
//   k_build_hist    rank-hash of every k-mer (Horner, :56-73), histogram by key
//   (scan)          bucket offsets = exclusive scan of the histogram = the dense table offs[]
//   k_build_scatter positions into their buckets through per-key cursors (arrival order)
//   k_bucket_sort_* every bucket ascending — the order push_back yields at :160-167
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t kmer_hash_at(const uint8_t* __restrict__ text, uint64_t i, uint32_t k, uint32_t sigma)
{
    uint64_t h = 0;
    for (uint32_t j = 0; j < k; ++j) h = h * sigma + text[i + j];
    return h;
}

__global__ __launch_bounds__(KMX_BLOCK) void k_build_hist(const uint8_t* __restrict__ text, uint64_t npos, uint32_t k,
                                                          uint32_t sigma, uint32_t* __restrict__ hist)
{
    const uint64_t stride = uint64_t(gridDim.x) * KMX_BLOCK;
    for (uint64_t i = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x; i < npos; i += stride)
        atomicAdd(&hist[kmer_hash_at(text, i, k, sigma)], 1u);
}

// offs64 (exclusive scan of hist, n_keys + 1 entries) -> offs32 and the scatter cursors; largest bucket
__global__ __launch_bounds__(KMX_BLOCK) void k_build_offsets(const uint64_t* __restrict__ offs64, uint64_t n_keys,
                                                             uint32_t* __restrict__ offs32, uint32_t* __restrict__ cursor,
                                                             unsigned int* __restrict__ max_bucket)
{
    const uint64_t i = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    uint32_t sz = 0;
    if (i <= n_keys) {
        const uint32_t o = uint32_t(offs64[i]);
        offs32[i] = o;
        if (i < n_keys) { cursor[i] = o; sz = uint32_t(offs64[i + 1]) - o; }
    }
    for (int off = 32; off > 0; off >>= 1) sz = max(sz, uint32_t(__shfl_xor(int(sz), off)));
    if (lane_id() == 0 && sz) atomicMax(max_bucket, sz);
}

__global__ __launch_bounds__(KMX_BLOCK) void k_build_scatter(const uint8_t* __restrict__ text, uint64_t npos, uint32_t k,
                                                             uint32_t sigma, uint32_t* __restrict__ cursor,
                                                             uint32_t* __restrict__ positions)
{
    const uint64_t stride = uint64_t(gridDim.x) * KMX_BLOCK;
    for (uint64_t i = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x; i < npos; i += stride)
        positions[atomicAdd(&cursor[kmer_hash_at(text, i, k, sigma)], 1u)] = uint32_t(i);
}

// one wave per bucket of at most KMX_PSORT_CAP positions
__global__ __launch_bounds__(KMX_BLOCK) void k_bucket_sort_wave(const uint32_t* __restrict__ offs, uint64_t n_keys,
                                                                uint32_t* __restrict__ positions)
{
    __shared__ uint32_t buf[KMX_BLOCK / KMX_WAVE][KMX_PSORT_CAP];
    const uint32_t lane = lane_id(), wv = threadIdx.x / KMX_WAVE;
    const uint64_t wave = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    const uint64_t n_waves = uint64_t(gridDim.x) * (KMX_BLOCK / KMX_WAVE);
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (uint64_t h = wave; h < n_keys; h += n_waves) {
        const uint32_t lo = offs[h], len = offs[h + 1] - lo;
        if (len < 2 || len > KMX_PSORT_CAP) continue;                    // wave-uniform
        uint32_t n2 = 2;
        while (n2 < len) n2 <<= 1;
        uint32_t* __restrict__ seg = positions + lo;
        for (uint32_t t = lane; t < n2; t += KMX_WAVE) buf[wv][t] = t < len ? seg[t] : 0xFFFFFFFFu;
        wsync();
        bitonic_lds(buf[wv], n2, lane, KMX_WAVE, wsync);
        for (uint32_t t = lane; t < len; t += KMX_WAVE) seg[t] = buf[wv][t];
        wsync();
    }
}

// one 1024-thread block per bucket of KMX_PSORT_CAP+1 .. KMX_PSORT_BLOCK_CAP positions
__global__ __launch_bounds__(1024) void k_bucket_sort_block(const uint32_t* __restrict__ offs, uint64_t n_keys,
                                                            uint32_t* __restrict__ positions)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sbuf[];
    const uint32_t tid = threadIdx.x;
    for (uint64_t h = blockIdx.x; h < n_keys; h += gridDim.x) {
        const uint32_t lo = offs[h], len = offs[h + 1] - lo;
        if (len <= KMX_PSORT_CAP || len > KMX_PSORT_BLOCK_CAP) continue; // block-uniform
        uint32_t n2 = 2;
        while (n2 < len) n2 <<= 1;
        uint32_t* __restrict__ seg = positions + lo;
        for (uint32_t t = tid; t < n2; t += 1024) sbuf[t] = t < len ? seg[t] : 0xFFFFFFFFu;
        __syncthreads();
        bitonic_lds(sbuf, n2, tid, 1024u, [] { __syncthreads(); });
        for (uint32_t t = tid; t < len; t += 1024) seg[t] = sbuf[t];
        __syncthreads();
    }
}

void launch_bucket_sort_block(hipStream_t s, const uint32_t* d_offs, uint64_t n_keys, uint32_t* d_positions);

// padded[h] = bucket size rounded up to a 128-byte line (32 positions); also counts the non-empty buckets
__global__ __launch_bounds__(KMX_BLOCK) void k_build_padded(const uint32_t* __restrict__ offs, uint64_t n_keys,
                                                            uint32_t* __restrict__ padded, unsigned int* __restrict__ present)
{
    const uint64_t h = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    uint32_t c = 0;
    if (h < n_keys) {
        c = offs[h + 1] - offs[h];
        padded[h] = (c + 31u) & ~31u;
    }
    const uint64_t nz = __ballot(c != 0);
    if (lane_id() == 0 && nz) atomicAdd(present, (unsigned int)__popcll(nz));
}

__global__ __launch_bounds__(KMX_BLOCK) void k_build_aoffs(const uint64_t* __restrict__ pscan, uint64_t n_keys, uint32_t a0,
                                                           uint32_t* __restrict__ aoffs)
{
    const uint64_t h = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    if (h < n_keys) aoffs[h] = a0 + uint32_t(pscan[h]);
}

// dense table of the aligned copy: (start/32) << 5 | (count & 31); entry n_keys = end of the copy
__global__ __launch_bounds__(KMX_BLOCK) void k_build_atab(const uint32_t* __restrict__ offs, const uint32_t* __restrict__ aoffs,
                                                          uint64_t n_keys, uint32_t end, uint32_t* __restrict__ atab)
{
    const uint64_t h = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    if (h < n_keys) atab[h] = (aoffs[h] & ~31u) | ((offs[h + 1] - offs[h]) & 31u);
    else if (h == n_keys) atab[h] = end & ~31u;
}

// one wave per bucket: contiguous copy -> line-aligned copy (both inside the element's arena region)
__global__ __launch_bounds__(KMX_BLOCK) void k_build_aligned_copy(const uint32_t* __restrict__ offs,
                                                                  const uint32_t* __restrict__ aoffs, uint64_t n_keys,
                                                                  uint32_t* __restrict__ region)
{
    const uint32_t lane = lane_id();
    const uint64_t wave = (uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x) / KMX_WAVE;
    const uint64_t n_waves = uint64_t(gridDim.x) * (KMX_BLOCK / KMX_WAVE);
    for (uint64_t h = wave; h < n_keys; h += n_waves) {
        const uint32_t lo = offs[h], len = offs[h + 1] - lo, dst = aoffs[h];
        for (uint32_t t = lane; t < len; t += KMX_WAVE) region[dst + t] = region[lo + t];
    }
}

// Phase 1 of the device build of one element: histogram of the rank-hashes, bucket offsets (= the dense
// table), largest bucket, number of non-empty buckets and the size of the line-aligned copy.
// info (device, 4 words): [0] max bucket, [1] non-empty buckets; d_total[0] = npos, d_total[1] = padded total.
void launch_build_phase1(hipStream_t s, const uint8_t* d_text, uint64_t n, uint32_t k, uint32_t sigma, uint64_t n_keys,
                         uint32_t* d_hist, uint64_t* d_scratch_u64, uint64_t* d_bsum, uint32_t* d_offs, uint32_t* d_cursor,
                         unsigned int* d_info, unsigned long long* d_total)
{
    const uint64_t npos = n - k + 1;
    const unsigned int grid = (unsigned int)std::min<uint64_t>(blocks_for(npos, KMX_BLOCK), 256 * 64);
    (void)hipMemsetAsync(d_hist, 0, n_keys * sizeof(uint32_t), s);
    (void)hipMemsetAsync(d_info, 0, 4 * sizeof(unsigned int), s);
    hipLaunchKernelGGL(k_build_hist, dim3(grid), dim3(KMX_BLOCK), 0, s, d_text, npos, k, sigma, d_hist);
    launch_scan(s, d_hist, n_keys, d_bsum, d_scratch_u64, d_total);
    hipLaunchKernelGGL(k_build_offsets, dim3(blocks_for(n_keys + 1, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, d_scratch_u64, n_keys, d_offs,
                       d_cursor, d_info);
    // size of the aligned copy (d_hist is free again: reuse it for the padded counts)
    hipLaunchKernelGGL(k_build_padded, dim3(blocks_for(n_keys, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, d_offs, n_keys, d_hist, d_info + 1);
    launch_scan(s, d_hist, n_keys, d_bsum, d_scratch_u64, d_total + 1);
}

// Phase 2 (after the arena exists): positions into their buckets, every bucket ascending, and
// (d_aoffs != NULL) the line-aligned copy starting at element a0 of the region.  The scratch buffers are
// shared between elements, so cursors and the scan of the padded counts are recomputed here.
void launch_build_phase2(hipStream_t s, const uint8_t* d_text, uint64_t n, uint32_t k, uint32_t sigma, uint64_t n_keys,
                         const uint32_t* d_offs, uint32_t* d_hist, uint64_t* d_scratch_u64, uint64_t* d_bsum, uint32_t* d_cursor,
                         unsigned int* d_info, unsigned long long* d_total, uint32_t* d_region, uint32_t* d_aoffs, uint32_t a0,
                         int sort_mode, uint32_t* d_atab, uint32_t region_end)
{
    // sort_mode: 0 = scatter + wave sorts, 1 = + block sorts (a bucket beyond KMX_PSORT_CAP),
    //            2 = the positions already lie sorted in d_region (sort_kmer_positions: a bucket beyond the block sort)
    const uint64_t npos = n - k + 1;
    const unsigned int grid = (unsigned int)std::min<uint64_t>(blocks_for(npos, KMX_BLOCK), 256 * 64);
    const unsigned int wblocks = (unsigned int)std::min<uint64_t>((n_keys + 3) / 4, 256 * 32);
    if (sort_mode != 2) {
        (void)hipMemcpyAsync(d_cursor, d_offs, n_keys * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
        hipLaunchKernelGGL(k_build_scatter, dim3(grid), dim3(KMX_BLOCK), 0, s, d_text, npos, k, sigma, d_cursor, d_region);
        hipLaunchKernelGGL(k_bucket_sort_wave, dim3(wblocks ? wblocks : 1), dim3(KMX_BLOCK), 0, s, d_offs, n_keys, d_region);
        if (sort_mode == 1) launch_bucket_sort_block(s, d_offs, n_keys, d_region);
    }
    if (d_aoffs) {
        hipLaunchKernelGGL(k_build_padded, dim3(blocks_for(n_keys, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, d_offs, n_keys, d_hist, d_info + 1);
        launch_scan(s, d_hist, n_keys, d_bsum, d_scratch_u64, d_total + 1);
        hipLaunchKernelGGL(k_build_aoffs, dim3(blocks_for(n_keys, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, d_scratch_u64, n_keys, a0, d_aoffs);
        hipLaunchKernelGGL(k_build_aligned_copy, dim3(wblocks ? wblocks : 1), dim3(KMX_BLOCK), 0, s, d_offs, d_aoffs, n_keys, d_region);
        if (d_atab)
            hipLaunchKernelGGL(k_build_atab, dim3(blocks_for(n_keys + 1, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, d_offs, d_aoffs, n_keys, region_end, d_atab);
    }
}

// directory over the sorted distinct keys of an open table: dir[j] = first index with ukeys[index] >= j << shift
__global__ __launch_bounds__(KMX_BLOCK) void k_build_dir(const uint64_t* __restrict__ ukeys, uint64_t n_ukeys, uint32_t shift,
                                                         uint32_t n_dir, uint32_t* __restrict__ dir)
{
    const uint64_t j = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    if (j > n_dir) return;
    dir[j] = j == n_dir ? uint32_t(n_ukeys) : uint32_t(lower_bound_dev<uint64_t>(ukeys, n_ukeys, j << shift));
}

// cells: slot j of key h <- the j-th position of the key's group (0 beyond its size, and for a key whose group does not fit)
__global__ __launch_bounds__(KMX_BLOCK) void k_build_cells(const uint32_t* __restrict__ offs, const uint32_t* __restrict__ region,
                                                           uint64_t n_keys, uint32_t cell_shift, uint32_t* __restrict__ cells,
                                                           uint8_t* __restrict__ cnt8)
{
    const uint64_t i = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    const uint64_t h = i >> cell_shift;
    if (h >= n_keys) return;
    const uint32_t j = uint32_t(i) & ((1u << cell_shift) - 1u);
    const uint32_t lo = offs[h], c = offs[h + 1] - lo;
    const bool fits = c <= (1u << cell_shift);
    cells[i] = (fits && j < c) ? region[lo + j] : 0u;
    if (j == 0) cnt8[h] = fits ? uint8_t(c) : uint8_t(255);
}

void launch_build_cells(hipStream_t s, const uint32_t* d_offs, const uint32_t* d_region, uint64_t n_keys, uint32_t cell_shift,
                        uint32_t* d_cells, uint8_t* d_cnt8)
{
    hipLaunchKernelGGL(k_build_cells, dim3(blocks_for(n_keys << cell_shift, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, d_offs, d_region, n_keys,
                       cell_shift, d_cells, d_cnt8);
}

void launch_build_dir(hipStream_t s, const uint64_t* d_ukeys, uint64_t n_ukeys, uint32_t shift, uint32_t n_dir, uint32_t* d_dir)
{
    hipLaunchKernelGGL(k_build_dir, dim3(blocks_for(uint64_t(n_dir) + 1, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, d_ukeys, n_ukeys, shift, n_dir, d_dir);
}

// second stage for buckets beyond the wave sort's capacity (call when max bucket > KMX_PSORT_CAP)
void launch_bucket_sort_block(hipStream_t s, const uint32_t* d_offs, uint64_t n_keys, uint32_t* d_positions)
{
    const size_t lds = size_t(KMX_PSORT_BLOCK_CAP) * 4;
    allow_big_lds(reinterpret_cast<const void*>(k_bucket_sort_block), lds, 1);
    const unsigned int blocks = (unsigned int)std::min<uint64_t>(n_keys, 256 * 4);
    hipLaunchKernelGGL(k_bucket_sort_block, dim3(blocks ? blocks : 1), dim3(1024), lds, s, d_offs, n_keys, d_positions);
}

void launch_prefix_sort_small(hipStream_t s, const KmxIndexDev* ix, const uint64_t* qoff, const QueryDesc& d, uint64_t n_prefix,
                              const uint64_t* hit_off, const uint32_t* arena, uint32_t* out)
{
    unsigned int blocks = (unsigned int)std::min<uint64_t>((n_prefix + 3) / 4, 256 * 32);
    hipLaunchKernelGGL(k_prefix_sort_small, dim3(blocks ? blocks : 1), dim3(KMX_BLOCK), 0, s, ix, qoff, d, n_prefix, hit_off, arena, out);
}

void launch_prefix_merge_small(hipStream_t s, const KmxIndexDev* ix, const uint64_t* qoff, const QueryDesc& d, uint64_t n_prefix,
                               const uint64_t* hit_off, const uint32_t* arena, uint32_t* out)
{
    unsigned int blocks = (unsigned int)std::min<uint64_t>((n_prefix + 3) / 4, 256 * 32);
    hipLaunchKernelGGL(k_prefix_merge_small, dim3(blocks ? blocks : 1), dim3(KMX_BLOCK), 0, s, ix, qoff, d, n_prefix, hit_off, arena, out);
}

void launch_prefix_merge_pass(hipStream_t s, const QueryDesc& d, uint64_t n_prefix, const uint64_t* tile_off, uint64_t max_tiles,
                              const uint64_t* hit_off, uint32_t* out, uint32_t* tmp, uint32_t pass)
{
    hipLaunchKernelGGL(k_prefix_merge_pass, dim3((unsigned int)std::max<uint64_t>(max_tiles, 1)), dim3(KMX_BLOCK), 0, s, d, n_prefix, tile_off,
                       hit_off, out, tmp, pass);
}

uint64_t prefix_merge_tile() { return KMX_PM_TILE; }

// ---------------------------------------------------------------------------
// Prefix levels (KmxElemDev::n_levels) are built by the engine itself: the batch of ALL m-mers, in rank-hash order, is
// searched like any other batch; its hit_off / positions are the level.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(KMX_BLOCK) void k_all_kmers(uint32_t m, uint32_t sigma, uint64_t nq, uint8_t* __restrict__ qranks,
                                                         uint64_t* __restrict__ qoff)
{
    const uint64_t h = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    if (h > nq) return;
    qoff[h] = h * m;
    if (h == nq) return;
    uint64_t v = h;
    for (uint32_t t = m; t-- > 0;) { qranks[h * m + t] = uint8_t(v % sigma); v /= sigma; }     // Horner's digits, last letter first
}

__global__ __launch_bounds__(KMX_BLOCK) void k_narrow_offsets(const uint64_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ out)
{
    const uint64_t i = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    if (i < n) out[i] = uint32_t(in[i]);
}

void launch_all_kmers(hipStream_t s, uint32_t m, uint32_t sigma, uint64_t nq, uint8_t* d_qranks, uint64_t* d_qoff)
{
    hipLaunchKernelGGL(k_all_kmers, dim3(blocks_for(nq + 1, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, m, sigma, nq, d_qranks, d_qoff);
}

// kmx_result_gather_device: the hit_off entries a part brought along count from the part's first hit; on the gathering device
// they are moved behind the hits of the parts in front
__global__ __launch_bounds__(KMX_BLOCK) void k_rebase_offsets(uint64_t* __restrict__ off, uint64_t n, uint64_t add)
{
    const uint64_t i = uint64_t(blockIdx.x) * KMX_BLOCK + threadIdx.x;
    if (i < n) off[i] += add;
}

void launch_rebase_offsets(hipStream_t s, uint64_t* d_off, uint64_t n, uint64_t add)
{
    if (n && add) hipLaunchKernelGGL(k_rebase_offsets, dim3(blocks_for(n, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, d_off, n, add);
}

void launch_narrow_offsets(hipStream_t s, const uint64_t* d_in, uint64_t n, uint32_t* d_out)
{
    hipLaunchKernelGGL(k_narrow_offsets, dim3(blocks_for(n, KMX_BLOCK)), dim3(KMX_BLOCK), 0, s, d_in, n, d_out);
}

} // namespace kmx
