/* kmx.h — C-ABI of the MI355X-native k-mer exact-match batch search engine.
 *
 * This is the drop-in boundary underneath the reference's C++ template surface
 * (Clemapfel/kmer_index).  The reference has no FFI of its own; every entry point
 * below names the reference interface it stands in for (file:line relative to the
 * reference checkout).  The C++ host mirror in include/kmer_index_amd/ keeps the
 * reference's class and function names and calls only these functions.
 *
 * Conventions
 *   - plain pointers and sizes only; no exceptions cross the boundary: every call
 *     returns a kmx_status and kmx_last_error() gives a thread-local message;
 *   - letters are passed as alphabet RANKS, one byte per letter, exactly what
 *     seqan3::to_rank yields for the reference's alphabet_t (kmer_index.hpp:59,128);
 *   - positions are uint32_t text offsets (position_t = uint32_t, kmer_index.hpp:575);
 *   - input buffers are borrowed for the duration of the call; results are owned by
 *     the library until kmx_result_free;
 *   - one kmx_index may be searched from several host threads at once (the
 *     reference's search() is const, kmer_index.hpp:505) provided each call uses
 *     its own result handle (and, for the device-buffer form, its own stream): the
 *     host-buffer form runs on a stream owned by the result, so concurrent calls
 *     overlap on the GPU instead of queueing behind one another;
 *   - the engine needs the HIP runtime and a gfx950 device: there is no CPU path.
 */
#ifndef KMX_H
#define KMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMX_VERSION 4
#define KMX_MAX_KS 32               /* number of k values one index may hold                      */
#define KMX_MAX_DEVICES 16          /* replicas of one index (one per GPU of a node)              */
#define KMX_QUERY_SIZE_RANGE 10000  /* kmer_index::_query_size_range, kmer_index.hpp:401          */
#define KMX_SUBK_FANOUT_LIMIT 10000000ull /* sigma^(k-m) guard, kmer_index.hpp:119                */

typedef enum kmx_status {
    KMX_OK = 0,
    KMX_ERR_INVALID_ARGUMENT = 1,  /* bad pointer / size / k (static_assert kmer_index.hpp:42-43) */
    KMX_ERR_HIP = 2,               /* HIP runtime error (message has the HIP error string)       */
    KMX_ERR_OUT_OF_MEMORY = 3,
    KMX_ERR_NO_DEVICE = 4,         /* no gfx950 device visible: the engine never falls back       */
    KMX_ERR_TOO_LARGE = 5          /* text too long for 32-bit positions (kmer_index.hpp:169-170) */
} kmx_status;

/* Per-query status (array returned by kmx_result_status).  QUERY_TOO_LONG and
 * SUBK_FANOUT are the two places where the reference throws std::invalid_argument
 * (kmer_index.hpp:507-509 and :119-122); EMPTY_QUERY is its assert at :195. */
typedef enum kmx_query_status {
    KMX_Q_OK = 0,
    KMX_Q_TOO_LONG = 1,
    KMX_Q_SUBK_FANOUT = 2,
    KMX_Q_EMPTY_QUERY = 3,
    KMX_Q_BAD_RANK = 4   /* a letter >= sigma: not representable in the reference's alphabet_t */
} kmx_query_status;

/* How a query was served (array returned by kmx_result_kinds). */
typedef enum kmx_query_kind {
    KMX_KIND_NONE = 0,     /* error status or a part missed: empty result (kmer_index.hpp:204,224,524)  */
    KMX_KIND_EXACT = 1,    /* one bucket, bitmask bypassed (kmer_index.hpp:198-205, :529-530)           */
    KMX_KIND_STITCH = 2,   /* candidates = first part's bucket + validity mask (:207-339, :532-555)     */
    KMX_KIND_PREFIX = 3    /* m < k: every k-mer with this prefix + last-kmer fix-up (:115-148, :342-345) */
} kmx_query_kind;

typedef enum kmx_table_kind {
    KMX_TABLE_AUTO = 0,    /* dense when sigma^k <= 4 * (n-k+1), else open addressing                   */
    KMX_TABLE_OPEN = 1,    /* open-addressing {key, offset, count} slots, linear probing, load <= 0.5   */
    KMX_TABLE_DENSE = 2    /* direct addressing: offsets[sigma^k + 1]                                   */
} kmx_table_kind;

typedef struct kmx_options {
    uint32_t struct_size;  /* = sizeof(kmx_options)                                                     */
    int32_t device;        /* HIP device ordinal; -1 = current device                                   */
    uint32_t table_kind;   /* kmx_table_kind                                                            */
    uint32_t n_threads;    /* host threads for the per-k flatten (kmer_index ctor's n_threads, :481)    */
    uint32_t query_size_range; /* 0 = KMX_QUERY_SIZE_RANGE (extend_query_size_range, :498-502)          */
    uint32_t keep_host_arena;  /* keep a host copy of the position arena (kmx_index_arena_host)         */
    uint32_t host_flatten;     /* 1 = build every element on host threads; 0 = on the device when the
                                  key space allows (sigma^k <= 2^26), host otherwise                     */
    uint32_t no_aligned_copy;  /* 1 = keep neither of the derived bucket layouts: the second, 128-byte-line-aligned copy of
                                  long buckets (avg >= 32 positions; up to ~1.2x the position array, exact lookups read ~13 %
                                  less with it) and the fixed-size cells + one-byte count table of short buckets (avg <= 24;
                                  up to 8x the position array, +8...17 % queries/s on DNA5 k=10 / protein k=5).  Both are
                                  also left out on their own when their offsets would not fit 32 bits (texts near 2^32
                                  letters): kmx_index_info's device_bytes tells what an index really holds.          */
    /* ---- since KMX_VERSION 2 (a caller compiled against version 1 passes the shorter struct_size and gets one replica) ---- */
    uint32_t n_devices;        /* 0 / 1: one replica on `device`.  N > 1: the index is built once on devices[0] and its flat
                                  image replicated (device-to-device copies) into the HBM of devices[1..N-1]; a host-buffer
                                  batch search then shards the queries contiguously over the replicas — replica r gets queries
                                  [nq*r/N, nq*(r+1)/N) — and returns ONE result whose views concatenate the shards in replica
                                  order, byte for byte what one device returns (batches of fewer than 256 queries per
                                  replica go to the first replica whole).  0 also reads the environment
                                  variable KMX_DEVICES ("all" or a comma-separated list of ordinals), so that a caller of
                                  kmer::make_kmer_index uses every GPU of the node without a code change.  An ordinal may be
                                  listed more than once (replicas then share a device: only useful for testing).           */
    int32_t devices[KMX_MAX_DEVICES];
    /* ---- since KMX_VERSION 3 ---- */
    int32_t prefix_levels;     /* Sub-k queries (m < k, get_position_for_all_kmer_with_prefix kmer_index.hpp:115-148 + the std::sort
                                  of kmer_index_result.hpp:258): per dense element, up to this many PREFIX LEVELS are derived when
                                  the index is installed — level L holds, for every (k-L)-mer, the merged ascending list of all its
                                  occurrences, so a query of k-L letters copies one list instead of merging sigma^L buckets per
                                  query, and shorter ones merge sigma^L times fewer lists.  Results are identical; each level costs
                                  one more copy of the position array (4 bytes x text length) and is left out when it does not
                                  fit.  0 = default (KMX_PREFIX_LEVELS in the environment, else 2), -1 = none, N = at most N
                                  (<= 3).  Levels are derived data: not part of the on-disk image, rebuilt by kmx_index_load.
                                  A caller compiled against KMX_VERSION 1 or 2 (shorter struct_size: it cannot say what it
                                  wants) gets none.  kmx_index_levels reports what was built, kmx_index_memory what it costs. */
} kmx_options;

/* kmx_search_batch flags */
#define KMX_SEARCH_DEFAULT 0u
#define KMX_SEARCH_KEEP_MASKS 1u   /* keep candidate runs + compressed_bitset mask words for STITCH queries */
#define KMX_SEARCH_COUNT_ONLY 2u   /* stop after hit_off (no position lists are materialised)              */
#define KMX_SEARCH_ASYNC 4u        /* device form only: return once the first half of the search is enqueued (lookup,
                                      scan, the steady-state fill) without waiting for the counters; the search is
                                      completed by whatever touches the result next (counts / view / masks / free / a
                                      new search into it, or kmx_index_free of its index, which completes every search
                                      still pending on the index before it releases anything).  d_qranks and d_qoff must
                                      stay alive until then.  Two results used in turn keep the GPU busy across batches. */
#define KMX_SEARCH_REFERENCE_PLAN 8u /* answer every query from the element the REFERENCE's planner names (choose_search_scheme,
                                      kmer_index.hpp:407-476).  By default a single-k query longer than its k is answered from
                                      the largest k of the index that fits it instead of the k that wastes the fewest letters
                                      (kmer_index.hpp:465-473): the buckets to intersect are shorter by sigma^(difference) and
                                      the position lists are the same — but which of KMX_KIND_NONE / KMX_KIND_STITCH a query
                                      WITHOUT hits reports may differ (an absent part of the larger k ends the lookup early).
                                      KMX_SEARCH_KEEP_MASKS implies this flag: candidate runs and mask words are the
                                      reference's result object.  kmx_plan always reports the reference's tables.           */

typedef struct kmx_index kmx_index;
typedef struct kmx_result kmx_result;

/* Per-kernel timing collected with HIP events on the caller's stream. */
#define KMX_N_KERNELS 16
typedef struct kmx_kernel_stat {
    const char* name;
    uint64_t launches;
    double total_ms;
} kmx_kernel_stat;

/* ---- construction: stands in for kmer::make_kmer_index<ks...>(text, n_threads)
 *      (kmer_index.hpp:569-579) and the kmer_index constructor (:480-496), i.e. one
 *      kmer_index_element::create per k (:154-179) plus choose_search_scheme (:407-476).
 *      `ranks` holds n letters as ranks < sigma.  Requires 0 < k < 64/log2(sigma) for
 *      every k (:42-43), n >= max k and n + max k - 1 < 2^32 (:169-170). */
kmx_status kmx_index_build(const uint8_t* ranks, uint64_t n, uint32_t sigma, const uint32_t* ks,
                           uint32_t n_ks, const kmx_options* opts, kmx_index** out);
void kmx_index_free(kmx_index* index);

/* On-disk image of the flattened index: build once, load many (the intent stated in the thesis,
 * thesis/content/02_implementation.tex:44-46; not implemented by the reference).  kmx_index_load validates
 * magic, version, every size field (the file must be exactly as long as its element table says — nothing is
 * allocated on the word of a header the file cannot back) and a checksum, and then the CONTENTS the kernels index
 * with — group boundaries monotone and ending at npos, the positions of every group strictly ascending and inside
 * the text, distinct keys strictly ascending and inside the key space, at most half of the open-addressing slots
 * occupied (a full table would make the probe loop spin), every slot naming exactly the group of its key, the
 * aligned copy restating the groups, the text tail inside the alphabet — before touching the device.  An image
 * that loads can be searched without harm whatever else it holds (tests/test_image_cpu.py, test_image_gpu.py:
 * mutation fuzz). */
kmx_status kmx_index_save(const kmx_index* index, const char* path);
kmx_status kmx_index_load(const char* path, const kmx_options* opts, kmx_index** out);

/* Introspection of the flattened index (sizes in bytes are device-resident bytes). */
kmx_status kmx_index_info(const kmx_index* index, uint64_t* n, uint32_t* sigma, uint32_t* n_ks,
                          uint32_t* ks /* KMX_MAX_KS */, uint32_t* table_kinds /* KMX_MAX_KS */,
                          uint64_t* device_bytes);

/* What the device memory of one replica (kmx_index_info's device_bytes) is made of, in bytes: the position arrays proper
 * (4 bytes per k-mer start and element — the reference's buckets), and the layouts DERIVED from them that an options field or
 * an environment variable turns off: the 128-byte-line-aligned copy of long buckets and the cells of short ones
 * (no_aligned_copy), the prefix levels (prefix_levels); `tables` is the rest (offset / slot / key tables, directories, padding,
 * planner tables, tail).  Any pointer may be NULL. */
kmx_status kmx_index_memory(const kmx_index* index, uint64_t* positions, uint64_t* aligned_copy, uint64_t* cells,
                            uint64_t* prefix_levels, uint64_t* tables);

/* kmer_index::extend_query_size_range(new_maximum) (kmer_index.hpp:498-502): rebuilds the planner
 * table for query lengths < new_maximum and installs it.  Must not run concurrently with a search
 * on the same index. */
kmx_status kmx_index_extend_query_size_range(kmx_index* index, uint32_t new_maximum);

/* Planner tables — kmer_index::_optimal_nk_sum / _use_multi_search_scheme
 * (kmer_index.hpp:404-405) as built by choose_search_scheme (:407-476).  Pure host
 * code; usable without a device.  nk_off has range+1 entries into nk_flat; returns
 * the number of flat entries through *n_flat (call with nk_flat = NULL to size). */
kmx_status kmx_plan(const uint32_t* ks, uint32_t n_ks, uint32_t range, uint8_t* use_multi,
                    uint32_t* nk_off, uint32_t* nk_flat, uint64_t cap, uint64_t* n_flat);

/* The ENGINE's choice of element per query length (see KMX_SEARCH_REFERENCE_PLAN): k_used[q], q < range, is the k whose
 * element answers a query of q letters that the reference plans on ONE k — the reference's k for q <= k, exact multiples that
 * have no larger k to go to, and lengths whose rest could reach the sub-k fan-out guard on either choice; otherwise the largest
 * k of the index that fits q.  0 for lengths the reference answers with its multi-k scheme (their summands are kmx_plan's) and
 * for q == 0.  Pure host code; what a search without KMX_SEARCH_KEEP_MASKS / KMX_SEARCH_REFERENCE_PLAN runs on. */
kmx_status kmx_plan_engine(const uint32_t* ks, uint32_t n_ks, uint32_t range, uint32_t sigma, uint32_t* k_used);

/* choose_best_k (choose_best_k.hpp:12-60): the n_k (<= 10) values of k the reference's heuristic recommends for a
 * set of query lengths — candidates {29,27,25,23,21,19,17,13,11,10}, 3 points for a length the candidate divides,
 * 4 - miss points for a miss of at most 3, best scores first.  Pure host code. */
kmx_status kmx_choose_best_k(const uint64_t* query_lengths, uint64_t n_lengths, uint32_t n_k, uint32_t* ks_out);

/* kmer::detail::fast_pow (fast_pow.hpp:46-93), including its "0 on exp >= 63" rule. */
uint64_t kmx_fast_pow(uint64_t base, uint8_t exp);

/* ---- search: stands in for kmer_index::search(std::vector<alphabet_t>&) const
 *      (kmer_index.hpp:505-558) applied to a BATCH of queries, followed by
 *      kmer_index_result::to_vector() (kmer_index_result.hpp:244-260) per query.
 *      qranks: the queries' letters as ranks, concatenated; qoff[nq+1]: start of each
 *      query in qranks (qoff[0] = 0; qranks may be NULL when no query has a letter).  Host-buffer form:
 *      copies the inputs to the device, runs the device form on a stream owned by the result, and
 *      leaves the result ready for kmx_result_view.  Small batches (up to 8192 queries that are mostly
 *      plain lookups; kmer_index::search(query) is a batch of one) take a latency path instead: ONE launch
 *      that reads the queries from and writes the complete result to page-locked host memory — no copies,
 *      one wait (about 17-30 us for one query).  Such a result lives in host memory; kmx_result_view_device
 *      on it runs the device form then (the index must still exist). */
kmx_status kmx_search_batch(const kmx_index* index, const uint8_t* qranks, const uint64_t* qoff,
                            uint64_t nq, uint32_t flags, kmx_result** out);
/*      A batch whose descriptors or hit lists do not fit the device in one pass (more than 2^25 queries, or an out-of-memory
 *      on the first attempt) is streamed through the device in chunks: every chunk's result is moved to host memory and the
 *      device buffers serve the next chunk, a chunk that still does not fit is halved.  The result then has one part per
 *      chunk (kmx_result_parts), its host views and counts are those of the whole batch, device views do not exist for it. */

/* Device-buffer form: d_qranks / d_qoff are device pointers already resident in HBM (on an index with several replicas: in
 * the HBM of any of its devices — the replica on the device that owns d_qranks serves the call),
 * `stream` is a hipStream_t (NULL = the default stream).  All kernels are enqueued on
 * `stream`; the call returns after the one host read-back it needs (per-kind counts
 * and the hit total, 64 bytes) and with the fill kernels enqueued.  Passing a result
 * from a previous call in *inout reuses its device buffers (no allocation in the
 * steady state). */
kmx_status kmx_search_batch_device(const kmx_index* index, const void* d_qranks, const void* d_qoff,
                                   uint64_t nq, uint32_t flags, void* stream, kmx_result** inout);

/* Result access.  Device views are valid after the call returns (in stream order);
 * host views copy to pinned host memory on first use and synchronise the stream.
 *   hit_off[nq+1]  : start of query q's hits in `positions` (uint64)
 *   positions[...] : per query the ascending list of text offsets where it occurs —
 *                    kmer_index_result::to_vector() (kmer_index_result.hpp:244-260)
 *   status[nq]     : kmx_query_status (uint8)
 *   kinds[nq]      : kmx_query_kind   (uint8) */
kmx_status kmx_result_counts(const kmx_result* r, uint64_t* nq, uint64_t* n_hits, uint64_t* n_exact,
                             uint64_t* n_stitch, uint64_t* n_prefix, uint64_t* n_error);
kmx_status kmx_result_view_device(const kmx_result* r, const uint64_t** d_hit_off,
                                  const uint32_t** d_positions, const uint8_t** d_status);
kmx_status kmx_result_view(kmx_result* r, const uint64_t** hit_off, const uint32_t** positions,
                           const uint8_t** status, const uint8_t** kinds);

/* Multi-device results.  A result of a host-buffer search on an index with N replicas has N parts, part p holding
 * the queries [q_begin, q_end) on `device`; kmx_result_view / kmx_result_masks / kmx_result_counts present the parts as one
 * result, kmx_result_view_device is refused for N > 1 (there is no single device to point into) — use the per-part device
 * views, whose hit_off is local to the part (hit_off[0] == 0).  Every other result has exactly one part. */
kmx_status kmx_result_parts(const kmx_result* r, uint32_t* n_parts);
kmx_status kmx_result_part_view_device(const kmx_result* r, uint32_t part, int32_t* device, uint64_t* q_begin, uint64_t* q_end,
                                       const uint64_t** d_hit_off, const uint32_t** d_positions, const uint8_t** d_status);
/* The devices an index is replicated on (devices[] holds KMX_MAX_DEVICES entries). */
kmx_status kmx_index_devices(const kmx_index* index, uint32_t* n_devices, int32_t* devices);

/* The exchange step of SURVEY 8e behind the C-ABI ("hit lists gathered ... over xGMI"): the parts of a multi-device result
 * gathered into ONE set of device arrays in the HBM of `dst_device` (any device of the node; normally the first replica's):
 * every part's hit lists go device to device (hipMemcpyPeerAsync on the part's own stream — all links at once, each behind
 * its part's last kernel) to the displacement the parts in front of it leave, its hit_off entries are rebased by that
 * displacement on the destination, its statuses follow.  What arrives is byte for byte what one device returns for the whole
 * batch: d_hit_off[nq + 1] (global), d_positions[n_hits], d_status[nq].  The arrays belong to the result (freed with it,
 * reused by the next gather into it) and are complete when the call returns.  A single-part result on dst_device is
 * returned where it lies (no copy).  Refused for chunk-streamed results (they live in host memory) and COUNT_ONLY
 * searches have no positions (d_positions = NULL). */
kmx_status kmx_result_gather_device(kmx_result* r, int32_t dst_device, const uint64_t** d_hit_off, const uint32_t** d_positions,
                                    const uint8_t** d_status);

/* KMX_SEARCH_KEEP_MASKS only — the reference's zero-copy result view
 * (kmer_index_result.hpp:15-24: pointers to bucket vectors + a compressed_bitset).
 * For a STITCH query q (kinds[q] == KMX_KIND_STITCH):
 *   cand_src[q]   : arena index of its candidate run = the first part's bucket
 *                   (kmer_index.hpp:272, :532), cand_count[q] ascending positions;
 *   mask_base[q]  : index of its first word in mask_words; it owns
 *                   cand_count[q]/64 + 1 words in compressed_bitset layout
 *                   (compressed_bitset.hpp:9-105: bit i = word i>>6, bit i&63);
 *                   bit i set <=> candidate i is a hit; padding bits are 0.
 * Entries of other queries are undefined.  The arena itself is reachable on the
 * host through kmx_index_arena_host when the index was built with keep_host_arena. */
kmx_status kmx_result_masks(kmx_result* r, const uint64_t** mask_base, const uint64_t** mask_words,
                            const uint32_t** cand_count, const uint64_t** cand_src);
kmx_status kmx_index_arena_host(const kmx_index* index, const uint32_t** arena, uint64_t* n_elems);

/* kmer_index_element<alphabet_t, position_t, k>::search_k(iterator) (kmer_index.hpp:183-190) = at(hash(it)) (:56-84): the
 * bucket of ONE k-mer, as a borrowed window into the index's host arena — no device round trip, nothing to free; valid as
 * long as the index.  `ranks` holds the k letters; *positions / *count receive the ascending text offsets of that k-mer,
 * NULL / 0 when the text does not hold it (the reference's nullptr).  Needs keep_host_arena; the element's offset table
 * (dense: 4 bytes per key; open addressing: 12 bytes per distinct key) is mirrored to host memory by the first call that
 * asks for that k.  Callable from several threads at once.  KMX_ERR_INVALID_ARGUMENT: no element for this k, a letter
 * outside the alphabet, an index without host arena. */
kmx_status kmx_index_bucket_host(const kmx_index* index, uint32_t k, const uint8_t* ranks, const uint32_t** positions,
                                 uint32_t* count);

/* Prefix levels actually built (kmx_options::prefix_levels asks, memory and the planner decide): levels[i] = number of levels
 * of the element of ks[i] (kmx_index_info's order), 0 for elements without.  levels holds KMX_MAX_KS entries. */
kmx_status kmx_index_levels(const kmx_index* index, uint32_t* levels);

void kmx_result_free(kmx_result* r);

/* Timing of the kernels launched for this index since the last reset (HIP events on
 * the stream each kernel ran on).  Enabled by kmx_stats_enable(index, 1). */
kmx_status kmx_stats_enable(kmx_index* index, int enable);
kmx_status kmx_stats_get(kmx_index* index, kmx_kernel_stat* stats /* KMX_N_KERNELS */, uint32_t* n);
kmx_status kmx_stats_reset(kmx_index* index);

/* Debug aid: 16 words of range-violation records written by -DKMX_CHECKED builds (word 0 = count;
 * always 0 in a normal build). */
kmx_status kmx_debug_words(const kmx_index* index, uint64_t* words16);

const char* kmx_last_error(void);
const char* kmx_status_string(kmx_status s);
uint32_t kmx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KMX_H */
