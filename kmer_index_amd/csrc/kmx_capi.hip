// C-ABI (include/kmx.h) over the gfx950 kernels: index upload, batch orchestration,
// result views, per-kernel HIP-event timing.  Host C++ only; every device-side step
// is a kernel from kmx_kernels.hip.  There is deliberately no CPU search path here:
// without a device every search entry point fails with KMX_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "kmx_host.h"
#include "kmx_kernels.h"

namespace {

thread_local std::string g_err;

kmx_status fail(kmx_status st, const std::string& msg)
{
    g_err = msg;
    return st;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            kmx_status st__ = (e__ == hipErrorOutOfMemory) ? KMX_ERR_OUT_OF_MEMORY : KMX_ERR_HIP; \
            if (e__ == hipErrorNoDevice || e__ == hipErrorInvalidDevice) st__ = KMX_ERR_NO_DEVICE; \
            return fail(st__, std::string(#expr) + ": " + hipGetErrorString(e__));             \
        }                                                                                      \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 16 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// Host side of a result: page-locked so that the D2H copies of kmx_result_view run at link speed
// (pageable malloc memory as the fallback when pinning fails).
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    bool ensure(size_t bytes)
    {
        if (bytes <= cap) return true;
        release();
        if (bytes >= (size_t(1) << 20) && hipHostMalloc(&p, bytes + 64, hipHostMallocDefault) == hipSuccess) {
            pinned = true;                                  // small views are not worth a pinning call
        } else {
            (void)hipGetLastError();
            p = malloc(bytes + 64);
            pinned = false;
        }
        cap = p ? bytes + 64 : 0;
        return p != nullptr;
    }
    bool ensure_pageable(size_t bytes)                      // plain malloc memory (chunk copies of a streamed batch: not worth pinning)
    {
        if (bytes <= cap && !pinned) return true;
        release();
        p = malloc(bytes + 64);
        cap = p ? bytes + 64 : 0;
        return p != nullptr;
    }
    bool ensure_pinned(size_t bytes)                        // page-locked or nothing
    {
        if (bytes <= cap && pinned) return true;
        release();
        if (hipHostMalloc(&p, bytes + 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
        pinned = true;
        cap = bytes + 64;
        return true;
    }
    void release()
    {
        if (p) {
            if (pinned) (void)hipHostFree(p); else free(p);
        }
        p = nullptr; cap = 0; pinned = false;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

enum KernelId {
    K_LOOKUP = 0, K_SCAN, K_PARTITION, K_FILL, K_VALIDATE, K_COMPACT, K_PREFIX_LEN, K_MERGE_PASS,
    K_PREFIX_SORT_SMALL, K_PREFIX_MERGE_SMALL, K_PREFIX_SORT_BLOCK, K_SMALL, K_PREFIX_BANDS, K_PREFIX_SPLIT, K_COUNT
};
const char* const kKernelNames[K_COUNT] = {
    "k_lookup", "k_scan", "k_partition", "k_fill", "k_validate", "k_compact",
    "k_prefix_len", "k_prefix_merge_pass", "k_prefix_sort_small", "k_prefix_merge_small", "k_prefix_sort_block", "k_small", "k_prefix_bands", "k_prefix_split"};

struct Stats {
    bool enabled = false;
    std::mutex mu;
    struct Pending { int id; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    uint64_t launches[K_COUNT] = {};
    double ms[K_COUNT] = {};

    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    // resolve every finished pair (blocks on unfinished ones)
    void drain()
    {
        for (auto& p : pending) {
            (void)hipEventSynchronize(p.b);
            float t = 0.f;
            if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) { ms[p.id] += t; launches[p.id] += 1; }
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        pending.clear();
    }
    void destroy()
    {
        drain();
        for (auto e : pool) (void)hipEventDestroy(e);
        pool.clear();
    }
};

} // namespace

// Results released with kmx_result_free wait here for the next search that arrives without a result of its
// own: a fresh result costs some twenty device allocations, which dominates the latency of small batches
// (kmer_index::search(query) is a batch of one).  Shared between the index and its results, so a result may
// outlive its index: a search still pending on a result (KMX_SEARCH_ASYNC) is listed here and completed by
// kmx_index_free before the index releases anything the second half would touch.
struct ResultPool {
    std::mutex mu;
    std::vector<kmx_result*> idle;
    std::vector<kmx_result*> pending;    // KMX_SEARCH_ASYNC searches whose second half has not run yet: kmx_index_free completes them
    bool closed = false;
};
constexpr size_t kPoolMaxResults = 8;
constexpr size_t kPoolMaxBytes = size_t(256) << 20;     // larger results give their memory back at once
constexpr size_t kSmallView = size_t(256) << 10;        // host views up to this size: one page-locked block, one wait

struct kmx_index {
    std::shared_ptr<ResultPool> pool = std::make_shared<ResultPool>();
    int device = 0;
    uint64_t n = 0;
    uint32_t sigma = 0;
    uint32_t range = KMX_QUERY_SIZE_RANGE;
    std::vector<uint32_t> ks;
    std::vector<uint32_t> table_kinds;
    std::vector<void*> allocs;          // device allocations owned by the index
    uint64_t device_bytes = 0;
    uint64_t bytes_positions = 0, bytes_aligned = 0, bytes_cells = 0, bytes_levels = 0;   // parts of device_bytes (kmx_index_memory)
    KmxIndexDev* d_index = nullptr;     // device copy of the header
    KmxIndexDev* d_index_fast = nullptr; // the same header around the engine's own planner table (kmx::make_fast_plan_entries): every search but KEEP_MASKS ones
    const KmxPlanEntry* d_plan_fast = nullptr;
    const uint32_t* d_arena = nullptr;  // the position arena (also in the header; passed to kernels directly)
    unsigned long long* d_dbg = nullptr; // KMX_CHECKED violation records
    struct ElemSizes { size_t n_offs, n_slots, n_ukeys, n_aoffs; };
    std::vector<ElemSizes> elem_sizes;   // element array lengths (for kmx_index_save)
    std::vector<uint8_t> tail;           // last kmax letters of the text
    KmxIndexDev h_header{};              // host copy of the device header (holds device pointers)
    kmx::FillVariant fill_variant{12, true};   // 3072-slot tiles (12 gathers in flight per thread), non-temporal stores
    bool rec32 = true;                   // every arena index fits 31 bits
    bool tiny_cells = false;             // some element has cells and at most four positions per key on average (k_lookup: 8 queries per thread)
    bool broken = false;                 // a failed kmx_index_extend_query_size_range left the replicas' planner tables inconsistent
    std::vector<uint32_t> host_arena;   // optional host mirror of the position arena
    Stats stats;
    // kmx_index_bucket_host: host mirror of one element's offset table (dense: offs; open: the sorted distinct keys + offs),
    // downloaded by the first call that asks for that k
    struct HostDir {
        std::once_flag once;
        kmx_status st = KMX_OK;
        std::string err;
        std::vector<uint32_t> offs;
        std::vector<uint64_t> ukeys;
    };
    std::unique_ptr<HostDir[]> host_dirs;
    std::vector<kmx_index*> peers;      // replicas 1..N-1 of a multi-device index (owned by replica 0, which is this object)
    size_t n_replicas() const { return 1 + peers.size(); }
    kmx_index* replica(size_t i) { return i ? peers[i - 1] : this; }
};

// What search_finish needs from the call that enqueued the first half of a search.
struct SearchCtx {
    kmx_index* ix = nullptr;
    const uint8_t* qr = nullptr;
    const uint64_t* qo = nullptr;
    hipStream_t s = nullptr;
    uint64_t tile_cap = 0, spec_tiles = 0;
    bool spec_fill = false;
    bool pending = false;
};

struct kmx_result {
    const kmx_index* index = nullptr;   // the index of the last search; dereferenced only while that search is pending (ctx.pending),
                                        // and kmx_index_free completes pending searches before the index goes away
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t flags = 0;
    uint64_t nq = 0, n_hits = 0, n_exact = 0, n_stitch = 0, n_prefix = 0, n_error = 0, n_none = 0;
    uint64_t n_mask_words = 0;
    // device
    DevBuf src, cnt, c0, aux, key, p1, kind, status, stitch_list, prefix_list, short_list, hit_off, bsum, ctr, tile_q, out,
        mask_words, stitch_hits, plen, poff, ptmp, pitems, pbands, pcuts, pbanded, psplits, ptiles, pscnt, pscratch, pmid, in_qranks, in_qoff;
    unsigned long long* h_ctr = nullptr;   // pinned
    // host mirrors
    HostBuf h_hit_off, h_positions, h_status, h_kinds, h_mask_base, h_mask_words, h_cand_count, h_cand_src, h_small;
    uint64_t* v_hit_off = nullptr; uint32_t* v_positions = nullptr; uint8_t* v_status = nullptr; uint8_t* v_kinds = nullptr;   // the current host view
    const uint64_t* m_base = nullptr; const uint64_t* m_words = nullptr; const uint32_t* m_ccnt = nullptr; const uint64_t* m_csrc = nullptr;   // ... and mask view
    // the latency path (k_small): queries in, complete result out through one page-locked block the kernel reads and writes
    uint32_t ctr_phase = 0;                // which of the two device counter blocks the current / last search counts into
    bool ctr_clean = false;                // the block the NEXT search will use is known to be zero (the last scan published and reset)
    HostBuf mailbox, small_in;
    DevBuf small_xchg;                     // the totals the workgroups of a multi-workgroup k_small launch exchange
    bool small_valid = false;              // the result of the last search lives in the mailbox only (no device buffers were written)
    bool host_valid = false, host_masks_valid = false;
    bool last_had_stitch = false;          // adaptive speculation: see kmx_search_batch_device
    bool last_had_long = false;            // ... or queries of very many parts (k_lookup_long)
    bool last_had_pairs = false;           // the previous batch held cross-referenced queries: k_lookup's variant (kmx_search_batch_device)
    std::shared_ptr<ResultPool> pool;      // where kmx_result_free parks this result (set by the search that made it)
    SearchCtx ctx;                         // the half-done search of a KMX_SEARCH_ASYNC call (search_finish completes it)
    hipEvent_t done = nullptr;             // recorded behind the first half's counter read-back
    bool quiesced = true;                  // no kernel of the last search can still be running on `stream`
    hipStream_t own_stream = nullptr;      // the stream of the host-buffer search form: one per result, so concurrent
                                           // kmx_search_batch calls on one index do not serialise behind a shared stream
    // a result over several replicas (host-buffer search on a multi-device index): the per-device results, in replica
    // order, and the first query of each (part_q0[n_parts] == nq).  The parent owns no device buffers.
    std::vector<kmx_result*> parts;
    std::vector<uint64_t> part_q0;
    std::vector<uint64_t> part_w0;         // first mask word of each part in the merged mask view
    // a batch too large for one pass (kmx_search_batch streams it through the device in chunks): the parts are host-resident
    // copies of the chunks' results, `worker` is the one result whose device buffers served every chunk
    bool chunked = false;                  // (parent) its parts are chunks
    bool host_chunk = false;               // (part) lives in host memory only
    kmx_result* worker = nullptr;
    kmx_result* worker2 = nullptr;         // ... and a second one: chunk i's results leave for the host while chunk i + 1 is searched
    // kmx_result_gather_device: the parts of a multi-device result gathered into one set of arrays on gather_device
    DevBuf g_hit_off, g_out, g_status;
    int gather_device = -1;
    hipStream_t gather_stream = nullptr;

    size_t device_bytes() const
    {
        size_t b = 0;
        for (const DevBuf* d : {&src, &cnt, &c0, &aux, &key, &p1, &kind, &status, &stitch_list, &prefix_list, &short_list, &hit_off, &bsum, &ctr,
                                &tile_q, &out, &mask_words, &stitch_hits, &plen, &poff, &ptmp, &pitems, &pbands, &pcuts, &pbanded, &psplits, &ptiles, &pscnt, &pscratch, &pmid, &in_qranks, &in_qoff, &small_xchg})
            b += d->cap;
        return b;
    }

    void release()
    {
        for (DevBuf* b : {&src, &cnt, &c0, &aux, &key, &p1, &kind, &status, &stitch_list, &prefix_list, &short_list, &hit_off, &bsum, &ctr,
                          &tile_q, &out, &mask_words, &stitch_hits, &plen, &poff, &ptmp, &pitems, &pbands, &pcuts, &pbanded, &psplits, &ptiles, &pscnt, &pscratch, &pmid, &in_qranks, &in_qoff, &small_xchg})
            b->release();
        for (HostBuf* b : {&h_hit_off, &h_positions, &h_status, &h_kinds, &h_mask_base, &h_mask_words, &h_cand_count, &h_cand_src, &h_small, &mailbox, &small_in})
            b->release();
        if (h_ctr) (void)hipHostFree(h_ctr);
        h_ctr = nullptr;
        if (done) (void)hipEventDestroy(done);
        done = nullptr;
        if (own_stream) (void)hipStreamDestroy(own_stream);
        own_stream = nullptr;
        if (gather_device >= 0) {
            int cur = 0;
            const bool have = hipGetDevice(&cur) == hipSuccess;
            (void)hipSetDevice(gather_device);
            g_hit_off.release(); g_out.release(); g_status.release();
            if (gather_stream) (void)hipStreamDestroy(gather_stream);
            gather_stream = nullptr;
            gather_device = -1;
            if (have) (void)hipSetDevice(cur);
        }
    }
};

namespace {

kmx_result* take_result(kmx_index* ix)
{
    kmx_result* r = nullptr;
    {
        std::lock_guard<std::mutex> lock(ix->pool->mu);
        if (!ix->pool->idle.empty()) { r = ix->pool->idle.back(); ix->pool->idle.pop_back(); }
    }
    if (!r) r = new kmx_result();
    r->pool = ix->pool;
    return r;
}

template <typename F>
void timed(kmx_index* ix, int id, hipStream_t s, F&& launch)
{
    if (!ix->stats.enabled) { launch(); return; }
    std::lock_guard<std::mutex> lock(ix->stats.mu);
    hipEvent_t a = ix->stats.get(), b = ix->stats.get();
    (void)hipEventRecord(a, s);
    launch();
    (void)hipEventRecord(b, s);
    ix->stats.pending.push_back({id, a, b});
}

template <typename T>
kmx_status upload(kmx_index* ix, const T* host, size_t count, const T** dev_out)
{
    void* p = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 16) + 16;   // +16: tables are probed with 16-byte loads
    HIP_TRY(hipMalloc(&p, bytes));
    ix->allocs.push_back(p);
    ix->device_bytes += bytes;
    if (count) HIP_TRY(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *dev_out = static_cast<const T*>(p);
    return KMX_OK;
}

// device scratch released at scope exit
struct DeviceScratch {
    std::vector<void*> ptrs;
    ~DeviceScratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T> hipError_t get(T** out, size_t count)
    {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, count * sizeof(T) + 64);
        if (e == hipSuccess) { ptrs.push_back(p); *out = static_cast<T*>(p); }
        return e;
    }
};

kmx_status check_device()
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(KMX_ERR_NO_DEVICE, "no HIP device visible: the kmx engine has no CPU search path");
    return KMX_OK;
}

} // namespace

// Cells (KmxElemDev::cnt8) for a dense element with short buckets: log2 of the cell size in positions, or 0 for "no cells".
// A cell must hold nearly every group — for i.i.d. text the sizes are Poisson(c) with c = npos / sigma^k, and a cell of
// 8 / 16 / 32 positions leaves about 1 / 4 / 5 % of the keys out at c = 3.5 / 10.5 / 24 (those take one more table read:
// measured, the smaller cell is as fast as the next larger one — DNA5 k = 10: 16 vs 32 positions, protein k = 5: 8 vs 16 —
// at two thirds of the memory) — and the cells must not dwarf the index.
static uint32_t cell_shift_for(uint32_t table_kind, uint64_t n_keys, uint64_t npos, uint64_t region, const kmx_options& o)
{
    if (table_kind != KMX_TABLE_DENSE || o.no_aligned_copy || region > npos) return 0;     // (long buckets have the aligned copy)
    if (const char* e = getenv("KMX_CELLS")) { if (!atoi(e)) return 0; }
    const double c = double(npos) / double(n_keys);
    uint32_t shift = c <= 3.5 ? 3u : c <= 10.5 ? 4u : c <= 24.0 ? 5u : 0u;
    if (const char* e = getenv("KMX_CELL_SHIFT")) { const int v = atoi(e); if (v >= 3 && v <= 5) shift = uint32_t(v); }   // tuning experiment
    if (!shift) return 0;
    if ((n_keys << shift) > 8 * npos) return 0;               // at most 8x the contiguous copy
    return shift;
}
static inline uint64_t up32_elems(uint64_t v) { return (v + 31) & ~uint64_t(31); }

// The two device copies of the header follow ix->h_header (d_index: the reference's planner table; d_index_fast: the engine's).
static hipError_t publish_headers(kmx_index* ix)
{
    hipError_t e = hipMemcpy(ix->d_index, &ix->h_header, sizeof(KmxIndexDev), hipMemcpyHostToDevice);
    if (e == hipSuccess && ix->d_index_fast) {
        KmxIndexDev hf = ix->h_header;
        if (ix->d_plan_fast) hf.plan = ix->d_plan_fast;
        e = hipMemcpy(ix->d_index_fast, &hf, sizeof hf, hipMemcpyHostToDevice);
    }
    return e;
}
// the header a search runs on
static inline const KmxIndexDev* header_for(const kmx_index* ix, uint32_t flags)
{
    static const bool ref_plan_only = getenv("KMX_REFERENCE_PLAN") != nullptr;      // (test / comparison knob)
    return ((flags & (KMX_SEARCH_KEEP_MASKS | KMX_SEARCH_REFERENCE_PLAN)) || !ix->d_index_fast || ref_plan_only) ? ix->d_index : ix->d_index_fast;
}

// Uploads flattened element images and everything around them (tail, planner table, header).
// Shared by kmx_index_build (images fresh from the flatten) and kmx_index_load (images from a file).
static kmx_status install_images_impl(std::vector<kmx::ElemImage>& images, const uint8_t* tail_kmax, uint64_t n, uint32_t sigma,
                                 uint32_t range, int device, const kmx_options& o, kmx_index** out, void* prebuilt_arena,
                                 uint64_t prebuilt_arena_elems = 0)
{
    auto* ix = new kmx_index();
    ix->device = device;
    ix->n = n;
    ix->sigma = sigma;
    ix->range = range;
    const uint32_t n_ks = uint32_t(images.size());
    uint32_t kmax = 0;
    for (auto& im : images) { ix->ks.push_back(im.k); kmax = std::max(kmax, im.k); }
    kmx_status st = KMX_OK;
    auto bail = [&](kmx_status s) { std::string keep = g_err; kmx_index_free(ix); g_err = keep; return s; };
    for (auto& im : images) {   // device-built dense tables
        if (im.d_offs_prebuilt) ix->allocs.push_back(const_cast<uint32_t*>(im.d_offs_prebuilt));
        if (im.d_atab_prebuilt) ix->allocs.push_back(const_cast<uint32_t*>(im.d_atab_prebuilt));
        if (im.d_ukeys_prebuilt) ix->allocs.push_back(const_cast<uint64_t*>(im.d_ukeys_prebuilt));
        if (im.d_slots_prebuilt) ix->allocs.push_back(const_cast<KmxSlot*>(im.d_slots_prebuilt));
    }

    KmxIndexDev h{};
    h.n = n; h.sigma = sigma; h.n_ks = n_ks; h.kmax = kmax; h.range = range;
    for (uint32_t j = 0; j < 64; ++j) {
        // fast_pow(sigma, j); products that leave 64 bits are never reached by a valid k
        // and are saturated so the fan-out guard (> 1e7) still fires
        unsigned __int128 v = 1;
        for (uint32_t t = 0; t < j; ++t) { v *= sigma; if (v > ~uint64_t(0)) { v = ~uint64_t(0); break; } }
        h.pw[j] = uint64_t(v);
    }
    // arena = every element's positions back to back
    uint64_t arena_elems = 0;
    for (auto& im : images) {
        if (!im.region) im.region = im.npos;
        if (im.region > im.npos) arena_elems = (arena_elems + 31) & ~uint64_t(31);   // an aligned copy needs a line-aligned base
        arena_elems += im.region;
    }
    // cells of the dense elements with short buckets follow the elements' regions (derived data: not in the image)
    std::vector<uint32_t> cell_shift(n_ks, 0);
    std::vector<uint64_t> cell_base(n_ks, 0);
    for (uint32_t i = 0; i < n_ks; ++i) {
        const auto& im = images[i];
        cell_shift[i] = cell_shift_for(im.table_kind, im.n_keys, im.npos, im.region, o);
        if (cell_shift[i] && (up32_elems(arena_elems) + (im.n_keys << cell_shift[i]) + 65536) * 4 >= (uint64_t(1) << 32)) cell_shift[i] = 0;   // keep 32-bit arena offsets
        if (!cell_shift[i]) continue;
        arena_elems = up32_elems(arena_elems);
        cell_base[i] = arena_elems;
        arena_elems += im.n_keys << cell_shift[i];
    }
    if (prebuilt_arena && prebuilt_arena_elems < arena_elems) {   // (cannot happen: both sides size it with cell_shift_for)
        fail(KMX_ERR_INVALID_ARGUMENT, "internal: the prebuilt arena is smaller than the index layout");
        return bail(KMX_ERR_INVALID_ARGUMENT);
    }
    {
        void* p = prebuilt_arena;                             // device-built elements already sit in it
        if (!p) {
            hipError_t e = hipMalloc(&p, arena_elems * 4 + KMX_ARENA_PAD);   // padded: k_fill reads 16 bytes at any element
            if (e != hipSuccess) { fail(KMX_ERR_OUT_OF_MEMORY, std::string("arena: ") + hipGetErrorString(e)); return bail(KMX_ERR_OUT_OF_MEMORY); }
        }
        ix->allocs.push_back(p);
        ix->device_bytes += arena_elems * 4;
        // the padding reads as 0xFFFFFFFF, k_fill's "do not store" (kmx_types.h, KMX_ARENA_PAD)
        {
            hipError_t e = hipMemset(static_cast<unsigned char*>(p) + arena_elems * 4, 0xFF, KMX_ARENA_PAD);
            if (e != hipSuccess) { fail(KMX_ERR_HIP, std::string("arena padding: ") + hipGetErrorString(e)); return bail(KMX_ERR_HIP); }
        }
        h.arena = static_cast<const uint32_t*>(p);
        ix->d_arena = h.arena;
    }
    h.arena_elems = arena_elems;
    ix->rec32 = (arena_elems + 65536) * 4 < (uint64_t(1) << 32);   // 32-bit byte offsets reach the whole arena
    if (const char* f64 = getenv("KMX_FORCE_REC64")) { if (atoi(f64)) ix->rec32 = false; }   // test hook: the >= 4 GiB arena path
    if (const char* fvs = getenv("KMX_FILL_VARIANT")) {
        // tuning knob: "<e>[n]", e.g. "8n", "16", "16n"
        int e = atoi(fvs);
        if (e == 4 || e == 8 || e == 12 || e == 16) ix->fill_variant = kmx::FillVariant{e, strchr(fvs, 'n') != nullptr};
    }
    {
        void* p = nullptr;
        if (hipMalloc(&p, 16 * 8) == hipSuccess) { (void)hipMemset(p, 0, 16 * 8); ix->allocs.push_back(p); h.dbg = static_cast<unsigned long long*>(p); ix->d_dbg = h.dbg; }
    }
    if (o.keep_host_arena) ix->host_arena.assign(arena_elems, 0);          // the mirror has the arena's layout, padding included
    uint64_t base = 0;
    for (uint32_t i = 0; i < n_ks; ++i) {
        auto& im = images[i];
        KmxElemDev& el = h.elems[i];
        el.k = im.k; el.table_kind = im.table_kind; el.log2cap = im.log2cap;
        el.n_ukeys = uint32_t(im.d_ukeys_prebuilt ? im.n_ukeys_prebuilt : im.ukeys.size());
        if (im.region > im.npos) base = (base + 31) & ~uint64_t(31);
        el.n_keys = im.n_keys; el.arena_base = base; el.npos = im.npos; el.region = im.region; el.atab = nullptr;
        ix->bytes_positions += im.npos * 4; ix->bytes_aligned += (im.region - im.npos) * 4;
        ix->table_kinds.push_back(im.table_kind);
        ix->elem_sizes.push_back({im.offs.size(), im.slots.size(), im.ukeys.size(),
                                  (im.d_atab_prebuilt || (im.table_kind == KMX_TABLE_DENSE && !im.atab.empty())) ? size_t(im.n_keys + 1) : size_t(0)});
        if (im.positions_on_device) {
            if (o.keep_host_arena) {
                hipError_t e = hipMemcpy(ix->host_arena.data() + base, h.arena + base, im.region * 4, hipMemcpyDeviceToHost);
                if (e != hipSuccess) { fail(KMX_ERR_HIP, std::string("arena download: ") + hipGetErrorString(e)); return bail(KMX_ERR_HIP); }
            }
        } else {
            hipError_t e = hipMemcpy(const_cast<uint32_t*>(h.arena) + base, im.positions.data(), im.region * 4, hipMemcpyHostToDevice);
            if (e != hipSuccess) { fail(KMX_ERR_HIP, std::string("arena upload: ") + hipGetErrorString(e)); return bail(KMX_ERR_HIP); }
            if (o.keep_host_arena) std::copy(im.positions.begin(), im.positions.begin() + im.region, ix->host_arena.begin() + base);
        }
        if (im.d_ukeys_prebuilt) {                              // open table built on the device
            const uint64_t nu = im.n_ukeys_prebuilt, cap = uint64_t(1) << im.log2cap;
            el.offs = im.d_offs_prebuilt; el.ukeys = im.d_ukeys_prebuilt; el.slots = im.d_slots_prebuilt;
            ix->device_bytes += (nu + 1) * 4 + nu * 8 + cap * sizeof(KmxSlot);
            ix->elem_sizes.back() = {size_t(nu + 1), size_t(cap), size_t(nu), size_t(0)};
        } else if (im.d_offs_prebuilt) {
            el.offs = im.d_offs_prebuilt;                       // (owned by ix->allocs since the top of this function)
            ix->device_bytes += (im.n_keys + 1) * 4;
            ix->elem_sizes.back().n_offs = im.n_keys + 1;
        } else if ((st = upload(ix, im.offs.data(), im.offs.size(), &el.offs)) != KMX_OK) return bail(st);
        if (im.d_atab_prebuilt) {
            el.atab = im.d_atab_prebuilt;
            ix->device_bytes += (im.n_keys + 1) * 4;
        } else if (im.table_kind == KMX_TABLE_DENSE && !im.atab.empty()) {
            if ((st = upload(ix, im.atab.data(), im.atab.size(), &el.atab)) != KMX_OK) return bail(st);
        }
        if (im.table_kind == KMX_TABLE_OPEN && !im.d_ukeys_prebuilt) {
            if ((st = upload(ix, im.slots.data(), im.slots.size(), &el.slots)) != KMX_OK) return bail(st);
            if ((st = upload(ix, im.ukeys.data(), im.ukeys.size(), &el.ukeys)) != KMX_OK) return bail(st);
        }
        el.dir = nullptr; el.dir_shift = 0; el.n_dir = 0;
        if (im.table_kind == KMX_TABLE_OPEN && el.n_ukeys >= 1024) {
            // directory over the sorted keys: 2^D cells of about 16 keys (at most 2^20 cells = 4 MB)
            uint32_t key_bits = 1;
            while (key_bits < 64 && (im.n_keys - 1) >> key_bits) ++key_bits;
            uint32_t lg = 0;
            while ((uint64_t(1) << lg) < el.n_ukeys) ++lg;
            const uint32_t D = std::min<uint32_t>(std::min<uint32_t>(20, key_bits), lg > 8 ? lg - 4 : 4);
            void* p = nullptr;
            if (hipMalloc(&p, ((size_t(1) << D) + 1) * 4 + 64) == hipSuccess) {
                ix->allocs.push_back(p);
                ix->device_bytes += ((size_t(1) << D) + 1) * 4;
                el.dir = static_cast<const uint32_t*>(p);
                el.dir_shift = key_bits - D;
                el.n_dir = 1u << D;
                kmx::launch_build_dir(nullptr, el.ukeys, el.n_ukeys, el.dir_shift, el.n_dir, static_cast<uint32_t*>(p));
                hipError_t le = hipGetLastError();
                if (le == hipSuccess) le = hipDeviceSynchronize();
                if (le != hipSuccess) { fail(KMX_ERR_HIP, std::string("key directory: ") + hipGetErrorString(le)); return bail(KMX_ERR_HIP); }
            } else {
                (void)hipGetLastError();                            // no directory: the searches run over the whole key array
            }
        }
        el.cnt8 = nullptr; el.cell_base = 0; el.cell_shift = 0; el.n_levels = 0;
        if (cell_shift[i]) {
            void* p = nullptr;
            hipError_t ce = hipMalloc(&p, im.n_keys + 64);                 // + 64: the lookup reads the aligned 16 bytes around an entry
            if (ce == hipSuccess) {
                ix->allocs.push_back(p);
                ix->device_bytes += im.n_keys;
                (void)hipMemset(p, 0, im.n_keys + 64);
                kmx::launch_build_cells(nullptr, el.offs, h.arena + base, im.n_keys, cell_shift[i], const_cast<uint32_t*>(h.arena) + cell_base[i],
                                        static_cast<uint8_t*>(p));
                ce = hipGetLastError();
                if (ce == hipSuccess) ce = hipDeviceSynchronize();
            }
            if (ce != hipSuccess) { fail(ce == hipErrorOutOfMemory ? KMX_ERR_OUT_OF_MEMORY : KMX_ERR_HIP, std::string("cells: ") + hipGetErrorString(ce)); return bail(KMX_ERR_HIP); }
            el.cnt8 = static_cast<const uint8_t*>(p);
            if (im.npos <= 4 * im.n_keys) ix->tiny_cells = true;
            ix->bytes_cells += ((im.n_keys << cell_shift[i]) * 4) + im.n_keys;
            el.cell_base = cell_base[i];
            el.cell_shift = cell_shift[i];
            if (o.keep_host_arena) {                                       // candidate runs of STITCH queries may lie in cells
                ce = hipMemcpy(ix->host_arena.data() + cell_base[i], h.arena + cell_base[i], (im.n_keys << cell_shift[i]) * 4, hipMemcpyDeviceToHost);
                if (ce != hipSuccess) { fail(KMX_ERR_HIP, std::string("cells download: ") + hipGetErrorString(ce)); return bail(KMX_ERR_HIP); }
            }
        }
        base += im.region;
        im = kmx::ElemImage();   // release host memory early
    }
    ix->host_dirs.reset(new kmx_index::HostDir[n_ks]);
    ix->tail.assign(tail_kmax, tail_kmax + kmax);
    if ((st = upload(ix, tail_kmax, kmax, &h.tail)) != KMX_OK) return bail(st);
    {
        std::vector<KmxPlanEntry> plan = kmx::make_plan_entries(ix->ks, range);
        if ((st = upload(ix, plan.data(), plan.size(), &h.plan)) != KMX_OK) return bail(st);
    }
    {
        const KmxIndexDev* d = nullptr;
        if ((st = upload(ix, &h, 1, &d)) != KMX_OK) return bail(st);
        ix->d_index = const_cast<KmxIndexDev*>(d);
        ix->h_header = h;
        // the engine's own planner table and a second header around it
        std::vector<KmxPlanEntry> fast = kmx::make_fast_plan_entries(ix->ks, range, sigma);
        if ((st = upload(ix, fast.data(), fast.size(), &ix->d_plan_fast)) != KMX_OK) return bail(st);
        if ((st = upload(ix, &h, 1, &d)) != KMX_OK) return bail(st);
        ix->d_index_fast = const_cast<KmxIndexDev*>(d);
        const hipError_t pe = publish_headers(ix);
        if (pe != hipSuccess) { fail(KMX_ERR_HIP, std::string("header: ") + hipGetErrorString(pe)); return bail(KMX_ERR_HIP); }
    }
    *out = ix;
    return KMX_OK;
}

// A second replica of `src` in the HBM of `device`: every array of the flat image copied device to device
// (hipMemcpyPeer: over xGMI between two GPUs of a node), the header rebuilt around the new pointers.
static kmx_status replicate_index(const kmx_index* src, int device, kmx_index** out)
{
    auto* ix = new kmx_index();
    ix->device = device;
    ix->n = src->n; ix->sigma = src->sigma; ix->range = src->range;
    ix->ks = src->ks; ix->table_kinds = src->table_kinds; ix->elem_sizes = src->elem_sizes; ix->tail = src->tail;
    ix->fill_variant = src->fill_variant; ix->rec32 = src->rec32; ix->tiny_cells = src->tiny_cells;
    ix->bytes_positions = src->bytes_positions; ix->bytes_aligned = src->bytes_aligned; ix->bytes_cells = src->bytes_cells; ix->bytes_levels = src->bytes_levels;
    auto bail = [&](kmx_status s) { std::string keep = g_err; kmx_index_free(ix); g_err = keep; return s; };
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { fail(KMX_ERR_NO_DEVICE, std::string("replica device: ") + hipGetErrorString(e)); return bail(KMX_ERR_NO_DEVICE); }
    kmx_status st = KMX_OK;
    auto clone = [&](const void* sp, size_t bytes, size_t pad, const void** dp) -> bool {
        *dp = nullptr;
        if (!sp) return true;
        void* p = nullptr;
        hipError_t e2 = hipMalloc(&p, std::max<size_t>(bytes, 16) + pad);
        if (e2 != hipSuccess) { st = fail(KMX_ERR_OUT_OF_MEMORY, std::string("replica: ") + hipGetErrorString(e2)); return false; }
        ix->allocs.push_back(p);
        ix->device_bytes += bytes;
        if (bytes) {
            e2 = hipMemcpyPeer(p, device, sp, src->device, bytes);
            if (e2 != hipSuccess) { st = fail(KMX_ERR_HIP, std::string("replica copy: ") + hipGetErrorString(e2)); return false; }
        }
        *dp = p;
        return true;
    };
    KmxIndexDev h = src->h_header;
    const void* p = nullptr;
    if (!clone(src->h_header.arena, h.arena_elems * 4 + KMX_ARENA_PAD, 0, &p)) return bail(st);
    h.arena = static_cast<const uint32_t*>(p); ix->d_arena = h.arena;
    if (!clone(src->h_header.tail, h.kmax, 16, &p)) return bail(st);
    h.tail = static_cast<const uint8_t*>(p);
    if (!clone(src->h_header.plan, size_t(h.range) * sizeof(KmxPlanEntry), 16, &p)) return bail(st);
    h.plan = static_cast<const KmxPlanEntry*>(p);
    if (!clone(src->d_plan_fast, src->d_plan_fast ? size_t(h.range) * sizeof(KmxPlanEntry) : 0, 16, &p)) return bail(st);
    ix->d_plan_fast = static_cast<const KmxPlanEntry*>(p);
    {
        void* d = nullptr;
        if (hipMalloc(&d, 16 * 8) == hipSuccess) { (void)hipMemset(d, 0, 16 * 8); ix->allocs.push_back(d); h.dbg = static_cast<unsigned long long*>(d); ix->d_dbg = h.dbg; }
        else { (void)hipGetLastError(); h.dbg = nullptr; }
    }
    for (uint32_t i = 0; i < h.n_ks; ++i) {
        const KmxElemDev& se = src->h_header.elems[i];
        KmxElemDev& de = h.elems[i];
        const auto& sz = src->elem_sizes[i];
        if (!clone(se.offs, sz.n_offs * 4, 16, &p)) return bail(st);
        de.offs = static_cast<const uint32_t*>(p);
        if (!clone(se.atab, sz.n_aoffs * 4, 16, &p)) return bail(st);
        de.atab = static_cast<const uint32_t*>(p);
        if (!clone(se.slots, sz.n_slots * sizeof(KmxSlot), 16, &p)) return bail(st);
        de.slots = static_cast<const KmxSlot*>(p);
        if (!clone(se.ukeys, sz.n_ukeys * 8, 16, &p)) return bail(st);
        de.ukeys = static_cast<const uint64_t*>(p);
        if (!clone(se.dir, se.dir ? (size_t(se.n_dir) + 1) * 4 : 0, 64, &p)) return bail(st);
        de.dir = static_cast<const uint32_t*>(p);
        if (!clone(se.cnt8, se.cnt8 ? size_t(se.n_keys) : 0, 64, &p)) return bail(st);
        de.cnt8 = static_cast<const uint8_t*>(p);
    }
    if (!clone(nullptr, 0, 0, &p)) return bail(st);
    {
        void* d = nullptr;
        e = hipMalloc(&d, sizeof(KmxIndexDev) + 16);
        if (e == hipSuccess) { ix->allocs.push_back(d); e = hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice); }
        if (e != hipSuccess) { fail(KMX_ERR_HIP, std::string("replica header: ") + hipGetErrorString(e)); return bail(KMX_ERR_HIP); }
        ix->d_index = static_cast<KmxIndexDev*>(d);
        ix->h_header = h;
        if (ix->d_plan_fast) {
            void* d2 = nullptr;
            e = hipMalloc(&d2, sizeof(KmxIndexDev) + 16);
            if (e == hipSuccess) { ix->allocs.push_back(d2); ix->d_index_fast = static_cast<KmxIndexDev*>(d2); e = publish_headers(ix); }
            if (e != hipSuccess) { fail(KMX_ERR_HIP, std::string("replica header: ") + hipGetErrorString(e)); return bail(KMX_ERR_HIP); }
        }
    }
    *out = ix;
    return KMX_OK;
}

// Prefix levels (KmxElemDev::n_levels, kmx_options::prefix_levels) of a freshly installed index: for every dense element and
// L = 1, 2, ... as long as the planner sends queries of k - L letters to that element, the batch of ALL (k - L)-mers is searched
// by the engine itself (level L is merged from level L - 1: sigma lists per query) and its hit_off / positions move behind the
// arena as the level's table and lists — one more copy of the n positions per level, bought for sub-k queries that copy one
// list instead of merging sigma^L buckets.  Optional by nature: a level that does not fit (device memory, 32-bit arena
// offsets, a key space too large to enumerate) is left out and so are the levels behind it.
static kmx_status search_finish(kmx_result* r);
static kmx_status add_prefix_levels_impl(kmx_index* ix, const kmx_options& o);
static kmx_status add_prefix_levels(kmx_index* ix, const kmx_options& o)
{
    // The level searches write their hit lists with ordinary stores: the lists are read right back (into the arena), where the
    // searches of a user stream them out with non-temporal ones.  (It also keeps the construction's k_fill launches apart from
    // the searches' in a kernel trace: another instantiation of the kernel.)
    const kmx::FillVariant keep = ix->fill_variant;
    ix->fill_variant.nt = false;
    const kmx_status st = add_prefix_levels_impl(ix, o);
    ix->fill_variant = keep;
    return st;
}
static kmx_status add_prefix_levels_impl(kmx_index* ix, const kmx_options& o)
{
    int want = o.prefix_levels;
    if (want == 0) {
        want = KMX_DEFAULT_PREFIX_LEVELS;
        if (const char* env = getenv("KMX_PREFIX_LEVELS")) want = atoi(env);
    }
    if (want <= 0) return KMX_OK;
    want = std::min(want, int(KMX_MAX_LEVELS));
    HIP_TRY(hipSetDevice(ix->device));
    const std::vector<KmxPlanEntry> plan = kmx::make_plan_entries(ix->ks, ix->range);
    auto up32w = [](uint64_t v) { return (v + 31) & ~uint64_t(31); };
    for (uint32_t e = 0; e < ix->h_header.n_ks; ++e) {
        for (int L = 1; L <= want; ++L) {
            KmxIndexDev& h = ix->h_header;
            KmxElemDev& el = h.elems[e];
            if (el.table_kind != KMX_TABLE_DENSE || uint32_t(L) >= el.k) break;
            const uint32_t m = el.k - uint32_t(L);
            if (m >= ix->range || plan[m].scheme != KMX_SCHEME_SINGLE || plan[m].elem != e) break;   // another element answers m
            const uint64_t nq = h.pw[m];                               // sigma^m lists
            const uint64_t n_pos = h.n - m + 1;                        // every position starts exactly one m-mer
            if (nq > (uint64_t(1) << 26) || nq * m > (uint64_t(1) << 30)) break;
            const uint64_t offs_at = up32w(h.arena_elems), base = up32w(offs_at + nq + 1), new_elems = base + n_pos;
            if ((new_elems + 65536) * 4 >= (uint64_t(1) << 32)) break;                 // 32-bit arena offsets (k_fill's records)
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); break; }
            // the batch's result (positions + descriptors), the queries and the grown arena next to the old one
            if (double(free_b) < 1.25 * double(new_elems * 4 + n_pos * 4 + nq * (m + 96)) + double(size_t(256) << 20)) break;
            void *d_q = nullptr, *d_off = nullptr, *grown = nullptr;
            kmx_result* r = nullptr;
            auto cleanup = [&] {
                if (d_q) (void)hipFree(d_q);
                if (d_off) (void)hipFree(d_off);
                if (grown) (void)hipFree(grown);
                if (r) kmx_result_free(r);
            };
            auto soft_fail = [&] { (void)hipGetLastError(); cleanup(); };             // the level is optional
            if (hipMalloc(&d_q, nq * m + 64) != hipSuccess || hipMalloc(&d_off, (nq + 1) * 8) != hipSuccess) { soft_fail(); break; }
            kmx::launch_all_kmers(nullptr, m, h.sigma, nq, static_cast<uint8_t*>(d_q), static_cast<uint64_t*>(d_off));
            if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { soft_fail(); break; }
            kmx_status st = kmx_search_batch_device(ix, d_q, d_off, nq, KMX_SEARCH_DEFAULT, nullptr, &r);
            if (st == KMX_ERR_OUT_OF_MEMORY) { soft_fail(); break; }
            if (st != KMX_OK) { cleanup(); return st; }
            if (r->n_hits != n_pos || r->n_error) {
                cleanup();
                return fail(KMX_ERR_HIP, "prefix level: the lists of all " + std::to_string(m) + "-mers do not add up to the text's positions");
            }
            if (hipMalloc(&grown, new_elems * 4 + KMX_ARENA_PAD) != hipSuccess) { soft_fail(); break; }
            uint32_t* g = static_cast<uint32_t*>(grown);
            hipError_t he = hipMemcpy(g, h.arena, h.arena_elems * 4, hipMemcpyDeviceToDevice);
            if (he == hipSuccess) he = hipMemset(g + h.arena_elems, 0, (base - h.arena_elems) * 4);
            if (he == hipSuccess) {
                kmx::launch_narrow_offsets(nullptr, r->hit_off.as<uint64_t>(), nq + 1, g + offs_at);
                he = hipGetLastError();
            }
            if (he == hipSuccess) he = hipMemcpy(g + base, r->out.as<uint32_t>(), n_pos * 4, hipMemcpyDeviceToDevice);
            if (he == hipSuccess) he = hipMemset(g + new_elems, 0xFF, KMX_ARENA_PAD);
            if (he == hipSuccess) he = hipDeviceSynchronize();
            if (he != hipSuccess) { std::string msg = hipGetErrorString(he); cleanup(); return fail(KMX_ERR_HIP, "prefix level: " + msg); }
            kmx_result_free(r); r = nullptr;
            // swap the arenas and publish the level
            void* old = const_cast<uint32_t*>(h.arena);
            ix->allocs.erase(std::remove(ix->allocs.begin(), ix->allocs.end(), old), ix->allocs.end());
            ix->allocs.push_back(grown);
            ix->device_bytes += (new_elems - h.arena_elems) * 4;
            ix->bytes_levels += (new_elems - h.arena_elems) * 4;
            h.arena = g; ix->d_arena = g;
            h.arena_elems = new_elems;
            el.lvl_offs_at[L - 1] = offs_at;
            el.lvl_base[L - 1] = base;
            el.n_levels = uint32_t(L);
            grown = nullptr;
            he = publish_headers(ix);
            if (he == hipSuccess) he = hipDeviceSynchronize();
            (void)hipFree(old);
            cleanup();
            if (he != hipSuccess) return fail(KMX_ERR_HIP, std::string("prefix level header: ") + hipGetErrorString(he));
        }
    }
    // the results the level searches parked in the pool hold device memory of the size of the index: give it back
    {
        std::vector<kmx_result*> idle;
        { std::lock_guard<std::mutex> lock(ix->pool->mu); idle.swap(ix->pool->idle); }
        for (kmx_result* r : idle) { r->release(); delete r; }
    }
    return KMX_OK;
}

// Resolves the replica set an options struct (or the KMX_DEVICES environment variable) asks for and clones the freshly
// built / loaded primary onto the other devices.  Leaves the caller's current device as it found it.
static kmx_status add_replicas(kmx_index* ix, const kmx_options& o)
{
    std::vector<int> devs;
    if (o.n_devices > 1) {
        if (o.n_devices > KMX_MAX_DEVICES) return fail(KMX_ERR_INVALID_ARGUMENT, "options.n_devices exceeds KMX_MAX_DEVICES");
        devs.assign(o.devices, o.devices + o.n_devices);
    } else if (o.n_devices == 0) {
        if (const char* env = getenv("KMX_DEVICES")) {
            int count = 0;
            (void)hipGetDeviceCount(&count);
            if (!strcmp(env, "all")) { for (int d = 0; d < count && d < KMX_MAX_DEVICES; ++d) devs.push_back(d); }
            else {
                for (const char* c = env; *c;) {
                    char* end = nullptr;
                    long v = strtol(c, &end, 10);
                    if (end == c) return fail(KMX_ERR_INVALID_ARGUMENT, "KMX_DEVICES: expected \"all\" or a comma-separated list of device ordinals");
                    devs.push_back(int(v));
                    c = (*end == ',') ? end + 1 : end;
                    if (*end && *end != ',') return fail(KMX_ERR_INVALID_ARGUMENT, "KMX_DEVICES: expected \"all\" or a comma-separated list of device ordinals");
                }
                if (devs.size() > KMX_MAX_DEVICES) return fail(KMX_ERR_INVALID_ARGUMENT, "KMX_DEVICES lists more than KMX_MAX_DEVICES devices");
            }
            // the primary already lives on options.device / the current device: it becomes (or stays) the first replica
            auto it = std::find(devs.begin(), devs.end(), ix->device);
            if (it != devs.end()) std::rotate(devs.begin(), it, it + 1);
            else if (!devs.empty()) devs.insert(devs.begin(), ix->device);
        }
    }
    if (devs.size() < 2) return KMX_OK;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    for (int d : devs)
        if (d < 0 || d >= count) return fail(KMX_ERR_NO_DEVICE, "a device of the replica list is not visible (" + std::to_string(d) + " of " + std::to_string(count) + ")");
    kmx_status st = KMX_OK;
    for (size_t i = 1; i < devs.size() && st == KMX_OK; ++i) {
        kmx_index* rep = nullptr;
        st = replicate_index(ix, devs[i], &rep);
        if (st == KMX_OK) { rep->pool = std::make_shared<ResultPool>(); ix->peers.push_back(rep); }
    }
    (void)hipSetDevice(ix->device);
    return st;
}

// kmx_options as this library version understands it, from a caller's struct of either version
static bool read_options(const kmx_options* opts, kmx_options& o)
{
    o = kmx_options{};
    o.struct_size = sizeof(kmx_options);
    o.device = -1;
    if (!opts) return true;
    const size_t v1 = offsetof(kmx_options, n_devices), v2 = offsetof(kmx_options, prefix_levels);
    if (opts->struct_size != sizeof(kmx_options) && opts->struct_size != v1 && opts->struct_size != v2) return false;
    memcpy(&o, opts, opts->struct_size);
    o.struct_size = sizeof(kmx_options);
    if (opts->struct_size == v1) o.n_devices = 1;          // a version-1 caller: one replica, no environment override
    if (opts->struct_size != sizeof(kmx_options)) o.prefix_levels = -1;   // a caller that cannot ask for prefix levels does not pay their HBM
    return true;
}


extern "C" {

const char* kmx_last_error(void) { return g_err.c_str(); }

const char* kmx_status_string(kmx_status s)
{
    switch (s) {
    case KMX_OK: return "ok";
    case KMX_ERR_INVALID_ARGUMENT: return "invalid argument";
    case KMX_ERR_HIP: return "HIP runtime error";
    case KMX_ERR_OUT_OF_MEMORY: return "out of device memory";
    case KMX_ERR_NO_DEVICE: return "no device";
    case KMX_ERR_TOO_LARGE: return "text too large for 32-bit positions";
    }
    return "unknown";
}

uint32_t kmx_version(void) { return KMX_VERSION; }

uint64_t kmx_fast_pow(uint64_t base, uint8_t exp) { return kmx::fast_pow(base, exp); }

kmx_status kmx_choose_best_k(const uint64_t* query_lengths, uint64_t n_lengths, uint32_t n_k, uint32_t* ks_out)
{
    if ((!query_lengths && n_lengths) || !ks_out || n_k == 0 || n_k > 10)
        return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_choose_best_k: need lengths, an output array and 1..10 values of k");
    const std::vector<uint32_t> ks = kmx::choose_best_k(query_lengths, n_lengths, n_k);
    for (size_t i = 0; i < ks.size(); ++i) ks_out[i] = ks[i];
    return KMX_OK;
}

kmx_status kmx_plan(const uint32_t* ks, uint32_t n_ks, uint32_t range, uint8_t* use_multi, uint32_t* nk_off,
                    uint32_t* nk_flat, uint64_t cap, uint64_t* n_flat)
{
    if (!ks || n_ks == 0 || n_ks > KMX_MAX_KS || range == 0) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_plan: bad ks / range");
    for (uint32_t i = 0; i < n_ks; ++i)
        if (ks[i] == 0) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_plan: k must be > 0");
    std::vector<uint32_t> v(ks, ks + n_ks);
    kmx::Plan p = kmx::make_plan(v, range);
    uint64_t total = 0;
    for (uint32_t q = 0; q < range; ++q) {
        if (use_multi) use_multi[q] = p.use_multi[q];
        if (nk_off) nk_off[q] = uint32_t(total);
        for (uint32_t k : p.nk_sum[q]) {
            if (nk_flat && total < cap) nk_flat[total] = k;
            ++total;
        }
    }
    if (nk_off) nk_off[range] = uint32_t(total);
    if (n_flat) *n_flat = total;
    return KMX_OK;
}

kmx_status kmx_plan_engine(const uint32_t* ks, uint32_t n_ks, uint32_t range, uint32_t sigma, uint32_t* k_used)
{
    if (!ks || !k_used || n_ks == 0 || n_ks > KMX_MAX_KS || range == 0 || sigma < 2) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_plan_engine: bad ks / range / sigma");
    for (uint32_t i = 0; i < n_ks; ++i)
        if (ks[i] == 0) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_plan_engine: k must be > 0");
    const std::vector<uint32_t> v(ks, ks + n_ks);
    const std::vector<KmxPlanEntry> fast = kmx::make_fast_plan_entries(v, range, sigma);
    for (uint32_t q = 0; q < range; ++q)
        k_used[q] = !q ? 0u : fast[q].scheme == KMX_SCHEME_SINGLE ? v[fast[q].elem]
                  : fast[q].scheme == KMX_SCHEME_REPLANNED ? v[fast[q].nparts >> KMX_PLAN_ALT_SHIFT] : 0u;
    return KMX_OK;
}

kmx_status kmx_index_build(const uint8_t* ranks, uint64_t n, uint32_t sigma, const uint32_t* ks, uint32_t n_ks,
                           const kmx_options* opts, kmx_index** out)
{
    if (!out) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_build: out is NULL");
    *out = nullptr;
    if (!ranks || !ks || n_ks == 0 || n_ks > KMX_MAX_KS)
        return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_build: need ranks and 1..KMX_MAX_KS values of k");
    kmx_options o;
    if (!read_options(opts, o)) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_build: options.struct_size mismatch");
    if (o.n_devices > 1) o.device = o.devices[0];
    uint32_t kmax = 0;
    for (uint32_t i = 0; i < n_ks; ++i) {
        if (!kmx::k_is_valid(sigma, ks[i]))
            return fail(KMX_ERR_INVALID_ARGUMENT,
                        "the hashspace for the current k cannot be represented with only a 64-bit integer. Please specify a valid k");
        for (uint32_t j = 0; j < i; ++j)
            if (ks[j] == ks[i]) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_build: duplicate k");
        kmax = std::max(kmax, ks[i]);
    }
    if (n < kmax) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_build: text shorter than the largest k");
    if (n + kmax - 1 >= 0xFFFFFFFFull) return fail(KMX_ERR_TOO_LARGE, "your text is too large for this configuration");
    const uint32_t range = o.query_size_range ? o.query_size_range : KMX_QUERY_SIZE_RANGE;
    if (range > 65535 * 9u) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_build: query_size_range too large");

    kmx_status st = check_device();
    if (st != KMX_OK) return st;
    int device = o.device;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    HIP_TRY(hipSetDevice(device));

    // letters must be ranks of the alphabet (the reference's alphabet_t cannot hold anything else)
    {
        uint32_t mx = 0;
        for (uint64_t i = 0; i < n; ++i) mx = ranks[i] > mx ? ranks[i] : mx;
        if (mx >= sigma) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_build: the text holds a letter outside the alphabet");
    }

    std::vector<kmx::ElemImage> images(n_ks);
    std::vector<char> on_device(n_ks, 0);

    // 1. elements whose key space fits a histogram are built on the device (k_build_*, k_bucket_sort_*)
    const bool aligned_copy = !o.no_aligned_copy && !(getenv("KMX_ALIGNED") && atoi(getenv("KMX_ALIGNED")) == 0);
    void* arena = nullptr;   // allocated by install_images_impl once every element's region is known
    DeviceScratch scratch;
    // histogram path: up to 2^30 keys (the dense-table limit) when the device has room for its scratch
    // (hist + scan + cursors + tables: ~40 B per key), 2^26 otherwise
    uint64_t DEVICE_BUILD_MAX_KEYS = uint64_t(1) << 26;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            while (DEVICE_BUILD_MAX_KEYS < (uint64_t(1) << 30) && (DEVICE_BUILD_MAX_KEYS << 1) * 40 + n * 24 < free_b / 2) DEVICE_BUILD_MAX_KEYS <<= 1;
        } else {
            (void)hipGetLastError();
        }
    }
    auto up32 = [](uint64_t v) { return (v + 31) & ~uint64_t(31); };
    uint64_t max_keys = 0;
    std::vector<char> sparse(n_ks, 0);   // key space beyond the histogram: sorted (hash, position) pairs, open table (kmx_build_sort.hip)
    bool any_sparse = false;
    for (uint32_t i = 0; i < n_ks; ++i) {
        const uint64_t nk = kmx::key_space(sigma, ks[i]);
        if (o.host_flatten) continue;
        const bool open = kmx::resolve_table_kind(sigma, ks[i], n, o.table_kind) == KMX_TABLE_OPEN;
        if (open && nk > (uint64_t(1) << 26)) { sparse[i] = 1; any_sparse = true; }      // (an open table of a small key space comes from the histogram too)
        else if (nk <= DEVICE_BUILD_MAX_KEYS) { on_device[i] = 1; max_keys = std::max(max_keys, nk); }
    }
    // per device-built element, filled by phase 1
    struct DevElem { uint32_t* d_offs = nullptr; uint32_t* d_aoffs = nullptr; uint32_t* d_atab = nullptr; uint32_t max_bucket = 0; uint64_t a0 = 0; };
    std::vector<DevElem> dev(n_ks);
    auto free_dev = [&] { for (auto& de : dev) { if (de.d_offs) (void)hipFree(de.d_offs); if (de.d_aoffs) (void)hipFree(de.d_aoffs); if (de.d_atab) (void)hipFree(de.d_atab); de = DevElem(); } };
    uint8_t* d_text = nullptr; uint32_t* d_hist = nullptr; uint64_t* d_scr = nullptr; uint64_t* d_bsum = nullptr; uint32_t* d_cursor = nullptr;
    unsigned int* d_info = nullptr; unsigned long long* d_total = nullptr;
    if (any_sparse && !max_keys) {
        hipError_t e = scratch.get(&d_text, n);
        if (e == hipSuccess) e = hipMemcpy(d_text, ranks, n, hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(KMX_ERR_OUT_OF_MEMORY, std::string("device build scratch: ") + hipGetErrorString(e));
    }
    if (max_keys) {
        hipError_t e = scratch.get(&d_text, n);
        if (e == hipSuccess) e = scratch.get(&d_hist, max_keys);
        if (e == hipSuccess) e = scratch.get(&d_scr, max_keys + 1);
        if (e == hipSuccess) e = scratch.get(&d_bsum, kmx::scan_blocks(max_keys));
        if (e == hipSuccess) e = scratch.get(&d_cursor, max_keys);
        if (e == hipSuccess) e = scratch.get(&d_info, 4);
        if (e == hipSuccess) e = scratch.get(&d_total, 2);
        if (e == hipSuccess) e = hipMemcpy(d_text, ranks, n, hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(KMX_ERR_OUT_OF_MEMORY, std::string("device build scratch: ") + hipGetErrorString(e));
        // phase 1: histogram, offsets (= the dense table), sizes
        for (uint32_t i = 0; i < n_ks; ++i) {
            if (!on_device[i]) continue;
            const uint64_t nk = kmx::key_space(sigma, ks[i]), npos = n - ks[i] + 1;
            e = hipMalloc(reinterpret_cast<void**>(&dev[i].d_offs), (nk + 1) * 4 + 64);
            if (e != hipSuccess) { free_dev(); return fail(KMX_ERR_OUT_OF_MEMORY, std::string("offs: ") + hipGetErrorString(e)); }
            kmx::launch_build_phase1(nullptr, d_text, n, ks[i], sigma, nk, d_hist, d_scr, d_bsum, dev[i].d_offs, d_cursor, d_info, d_total);
            e = hipGetLastError();                                  // a rejected launch is not reported by the copies below
            if (e != hipSuccess) { free_dev(); return fail(KMX_ERR_HIP, std::string("device build (launch): ") + hipGetErrorString(e)); }
            unsigned int info[4] = {0, 0, 0, 0};
            unsigned long long totals[2] = {0, 0};
            e = hipMemcpy(info, d_info, sizeof info, hipMemcpyDeviceToHost);                      // also synchronises
            if (e == hipSuccess) e = hipMemcpy(totals, d_total, sizeof totals, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { free_dev(); return fail(KMX_ERR_HIP, std::string("device build: ") + hipGetErrorString(e)); }
            kmx::ElemImage& im = images[i];
            // (a bucket too large for the LDS sorts — heavily repetitive text: phase 2 radix-sorts (hash, position) pairs instead)
            dev[i].max_bucket = info[0];
            im.k = ks[i]; im.n_keys = nk; im.npos = npos; im.positions_on_device = true;
            im.table_kind = kmx::resolve_table_kind(sigma, ks[i], n, o.table_kind);
            im.region = npos;
            const uint64_t present = info[1], padded_total = totals[1];
            if (aligned_copy && present && npos >= 32 * present && up32(npos) + padded_total < 0xFFFFFFFFull) {
                dev[i].a0 = up32(npos);
                im.region = dev[i].a0 + padded_total;
                e = hipMalloc(reinterpret_cast<void**>(&dev[i].d_aoffs), nk * 4 + 64);
                if (e == hipSuccess && im.table_kind == KMX_TABLE_DENSE) e = hipMalloc(reinterpret_cast<void**>(&dev[i].d_atab), (nk + 1) * 4 + 64);
                if (e != hipSuccess) { free_dev(); return fail(KMX_ERR_OUT_OF_MEMORY, std::string("aoffs: ") + hipGetErrorString(e)); }
            }
        }
    }

    // 2. the others on host threads (the constructor's thread pool, kmer_index.hpp:485-492)
    std::vector<std::string> errs(n_ks);
    std::vector<char> oks(n_ks, 1);
    {
        uint32_t T = std::max<uint32_t>(1, std::min<uint32_t>(o.n_threads ? o.n_threads : std::thread::hardware_concurrency(), n_ks));
        std::vector<std::thread> threads;
        std::atomic<uint32_t> next{0};
        for (uint32_t t = 0; t < T; ++t)
            threads.emplace_back([&] {
                for (;;) {
                    uint32_t i = next.fetch_add(1);
                    if (i >= n_ks) return;
                    if (!on_device[i] && !sparse[i]) oks[i] = kmx::flatten_element(ranks, n, sigma, ks[i], o.table_kind, images[i], errs[i], aligned_copy);
                }
            });
        for (auto& t : threads) t.join();
    }
    for (uint32_t i = 0; i < n_ks; ++i)
        if (!oks[i]) { free_dev(); return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_build: " + errs[i]); }

    for (uint32_t i = 0; i < n_ks; ++i)
        if (sparse[i]) {
            kmx::ElemImage& im = images[i];
            im.k = ks[i]; im.n_keys = kmx::key_space(sigma, ks[i]); im.npos = n - ks[i] + 1; im.region = im.npos;
            im.table_kind = KMX_TABLE_OPEN; im.positions_on_device = true;
        }

    // 3. every region is known now: allocate the arena, then phase 2 of the device-built elements
    uint64_t arena_elems = 0;
    std::vector<uint64_t> bases(n_ks);
    for (uint32_t i = 0; i < n_ks; ++i) {
        if (!images[i].region) images[i].region = images[i].npos;
        if (images[i].region > images[i].npos) arena_elems = up32(arena_elems);
        bases[i] = arena_elems;
        arena_elems += images[i].region;
    }
    for (uint32_t i = 0; i < n_ks; ++i) {                            // room for the cells install_images_impl derives (same rule there)
        uint32_t cs = cell_shift_for(images[i].table_kind, images[i].n_keys, images[i].npos, images[i].region, o);
        if (cs && (up32(arena_elems) + (images[i].n_keys << cs) + 65536) * 4 >= (uint64_t(1) << 32)) cs = 0;
        if (cs) arena_elems = up32(arena_elems) + (images[i].n_keys << cs);
    }
    {
        hipError_t e = hipMalloc(&arena, arena_elems * 4 + KMX_ARENA_PAD);      // padded: kernels read 16 bytes at any element
        if (e != hipSuccess) { free_dev(); return fail(KMX_ERR_OUT_OF_MEMORY, std::string("arena: ") + hipGetErrorString(e)); }
        // the padding between line-aligned groups is never written by the build kernels: defined contents (the image on
        // disk and the host mirror of the arena see it)
        e = hipMemset(arena, 0, arena_elems * 4 + KMX_ARENA_PAD);
        if (e != hipSuccess) { free_dev(); (void)hipFree(arena); return fail(KMX_ERR_HIP, std::string("arena: ") + hipGetErrorString(e)); }
    }
    auto free_sparse = [&] {
        for (auto& im : images) {
            if (im.d_ukeys_prebuilt) { (void)hipFree(const_cast<uint64_t*>(im.d_ukeys_prebuilt)); (void)hipFree(const_cast<uint32_t*>(im.d_offs_prebuilt));
                                       (void)hipFree(const_cast<KmxSlot*>(im.d_slots_prebuilt)); im.d_ukeys_prebuilt = nullptr; im.d_offs_prebuilt = nullptr; im.d_slots_prebuilt = nullptr; }
        }
    };
    for (uint32_t i = 0; i < n_ks; ++i) {
        if (!sparse[i]) continue;
        kmx::ElemImage& im = images[i];
        uint32_t key_bits = 1;
        while (key_bits < 64 && (im.n_keys - 1) >> key_bits) ++key_bits;
        kmx::SparseTables t;
        hipError_t e = kmx::build_sparse_element(nullptr, d_text, n, ks[i], sigma, key_bits, static_cast<uint32_t*>(arena) + bases[i], &t);
        if (e != hipSuccess) {
            free_dev(); free_sparse(); (void)hipFree(arena);
            return fail(e == hipErrorOutOfMemory ? KMX_ERR_OUT_OF_MEMORY : KMX_ERR_HIP, std::string("device build (sorted pairs): ") + hipGetErrorString(e));
        }
        im.d_ukeys_prebuilt = t.d_ukeys; im.d_offs_prebuilt = t.d_offs; im.d_slots_prebuilt = t.d_slots;
        im.n_ukeys_prebuilt = t.n_ukeys; im.log2cap = t.log2cap;
    }
    for (uint32_t i = 0; i < n_ks; ++i) {
        if (!on_device[i]) continue;
        kmx::ElemImage& im = images[i];
        uint32_t* d_region = static_cast<uint32_t*>(arena) + bases[i];
        int sort_mode = dev[i].max_bucket > KMX_PSORT_BLOCK_CAP ? 2 : dev[i].max_bucket > KMX_PSORT_CAP ? 1 : 0;
        hipError_t e = hipSuccess;
        if (sort_mode == 2) {
            uint32_t key_bits = 1;
            while (key_bits < 64 && (im.n_keys - 1) >> key_bits) ++key_bits;
            e = kmx::sort_kmer_positions(nullptr, d_text, n, ks[i], sigma, key_bits, d_region);
        }
        if (e == hipSuccess) {
            kmx::launch_build_phase2(nullptr, d_text, n, ks[i], sigma, im.n_keys, dev[i].d_offs, d_hist, d_scr, d_bsum, d_cursor, d_info, d_total,
                                     d_region, dev[i].d_aoffs, uint32_t(dev[i].a0), sort_mode, dev[i].d_atab, uint32_t(im.region));
            e = hipGetLastError();                                  // a rejected launch (e.g. the 128 KB dynamic-LDS block sort) would
            if (e == hipSuccess) e = hipDeviceSynchronize();        // leave buckets unsorted without any later call noticing
        }
        if (e != hipSuccess) { free_dev(); free_sparse(); (void)hipFree(arena); return fail(KMX_ERR_HIP, std::string("device build: ") + hipGetErrorString(e)); }
        if (im.table_kind == KMX_TABLE_DENSE) {
            im.d_offs_prebuilt = dev[i].d_offs;                      // the dense table itself
            im.d_atab_prebuilt = dev[i].d_atab;
            if (dev[i].d_aoffs) (void)hipFree(dev[i].d_aoffs);       // only needed to lay out the copy
            dev[i] = DevElem();
        } else {
            // open addressing requested: distinct keys / compact offsets / slots from the downloaded offsets
            std::vector<uint32_t> start(im.n_keys + 1), astart;
            e = hipMemcpy(start.data(), dev[i].d_offs, (im.n_keys + 1) * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess && dev[i].d_aoffs) {
                astart.resize(im.n_keys);
                e = hipMemcpy(astart.data(), dev[i].d_aoffs, im.n_keys * 4, hipMemcpyDeviceToHost);
            }
            if (e != hipSuccess) { free_dev(); free_sparse(); (void)hipFree(arena); return fail(KMX_ERR_HIP, std::string("offs download: ") + hipGetErrorString(e)); }
            im.offs.push_back(0);
            for (uint64_t j = 0; j < im.n_keys; ++j)
                if (start[j + 1] != start[j]) {
                    im.ukeys.push_back(j);
                    im.offs.push_back(start[j + 1]);
                    if (!astart.empty()) im.aoffs.push_back(astart[j]);
                }
            kmx::build_slots(im);
            im.aoffs.clear();                                        // only the slots carry it for open tables
            (void)hipFree(dev[i].d_offs);
            if (dev[i].d_aoffs) (void)hipFree(dev[i].d_aoffs);
            dev[i] = DevElem();
        }
    }

    st = install_images_impl(images, ranks + (n - kmax), n, sigma, range, device, o, out, arena, arena_elems);
    if (st != KMX_OK) return st;
    st = add_prefix_levels(*out, o);
    if (st == KMX_OK) st = add_replicas(*out, o);
    if (st != KMX_OK) { std::string keep = g_err; kmx_index_free(*out); *out = nullptr; g_err = keep; }
    return st;
}

static kmx_status search_finish(kmx_result* r);

void kmx_index_free(kmx_index* ix)
{
    if (!ix) return;
    for (kmx_index* peer : ix->peers) kmx_index_free(peer);
    ix->peers.clear();
    (void)hipSetDevice(ix->device);
    {
        // searches still pending on live results (KMX_SEARCH_ASYNC) would launch kernels on this index's memory when
        // their result is next touched: complete them now, while everything is still there
        std::vector<kmx_result*> pending;
        {
            std::lock_guard<std::mutex> lock(ix->pool->mu);
            pending = ix->pool->pending;
        }
        for (kmx_result* r : pending) (void)search_finish(r);
        std::vector<kmx_result*> idle;
        {
            std::lock_guard<std::mutex> lock(ix->pool->mu);
            ix->pool->closed = true;
            idle.swap(ix->pool->idle);
        }
        for (kmx_result* r : idle) { r->release(); delete r; }
    }
    (void)hipDeviceSynchronize();          // nothing of a completed search is still reading the image
    ix->stats.destroy();
    for (void* p : ix->allocs) (void)hipFree(p);
    delete ix;
}

kmx_status kmx_index_info(const kmx_index* ix, uint64_t* n, uint32_t* sigma, uint32_t* n_ks, uint32_t* ks,
                          uint32_t* table_kinds, uint64_t* device_bytes)
{
    if (!ix) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_info: index is NULL");
    if (n) *n = ix->n;
    if (sigma) *sigma = ix->sigma;
    if (n_ks) *n_ks = uint32_t(ix->ks.size());
    for (size_t i = 0; i < ix->ks.size(); ++i) {
        if (ks) ks[i] = ix->ks[i];
        if (table_kinds) table_kinds[i] = ix->table_kinds[i];
    }
    if (device_bytes) *device_bytes = ix->device_bytes;          // per replica
    return KMX_OK;
}

kmx_status kmx_index_memory(const kmx_index* ix, uint64_t* positions, uint64_t* aligned_copy, uint64_t* cells, uint64_t* prefix_levels,
                            uint64_t* tables)
{
    if (!ix) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_memory: index is NULL");
    if (positions) *positions = ix->bytes_positions;
    if (aligned_copy) *aligned_copy = ix->bytes_aligned;
    if (cells) *cells = ix->bytes_cells;
    if (prefix_levels) *prefix_levels = ix->bytes_levels;
    const uint64_t parts = ix->bytes_positions + ix->bytes_aligned + ix->bytes_cells + ix->bytes_levels;
    if (tables) *tables = ix->device_bytes > parts ? ix->device_bytes - parts : 0;
    return KMX_OK;
}

kmx_status kmx_index_devices(const kmx_index* ix, uint32_t* n_devices, int32_t* devices)
{
    if (!ix || !n_devices) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_devices: NULL argument");
    *n_devices = uint32_t(ix->n_replicas());
    if (devices)
        for (size_t i = 0; i < ix->n_replicas(); ++i) devices[i] = const_cast<kmx_index*>(ix)->replica(i)->device;
    return KMX_OK;
}

kmx_status kmx_index_extend_query_size_range(kmx_index* ix, uint32_t new_maximum)
{
    if (!ix || new_maximum == 0 || new_maximum > 65535 * 9u)
        return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_extend_query_size_range: bad argument");
    // Every replica gets the new planner table BEFORE any header says so: a step that fails (device memory on replica 2)
    // leaves all replicas on the old range and the old plan — shards of one batch never classify a length differently.
    // The caller's current device is left as it was found.
    int caller_dev = 0;
    const bool have_dev = hipGetDevice(&caller_dev) == hipSuccess;
    if (!have_dev) (void)hipGetLastError();
    auto restore = [&] { if (have_dev) (void)hipSetDevice(caller_dev); };
    const std::vector<KmxPlanEntry> plan = kmx::make_plan_entries(ix->ks, new_maximum);
    const std::vector<KmxPlanEntry> fast = kmx::make_fast_plan_entries(ix->ks, new_maximum, ix->sigma);
    std::vector<kmx_index*> all;
    all.push_back(ix);
    for (kmx_index* peer : ix->peers) all.push_back(peer);
    std::vector<const KmxPlanEntry*> d_plans(all.size(), nullptr), d_fast(all.size(), nullptr);
    for (size_t i = 0; i < all.size(); ++i) {
        hipError_t e = hipSetDevice(all[i]->device);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { restore(); return fail(KMX_ERR_HIP, std::string("kmx_index_extend_query_size_range: ") + hipGetErrorString(e)); }
        kmx_status st = upload(all[i], plan.data(), plan.size(), &d_plans[i]);     // (owned by the replica; unused if we stop here)
        if (st == KMX_OK && all[i]->d_index_fast) st = upload(all[i], fast.data(), fast.size(), &d_fast[i]);
        if (st != KMX_OK) { restore(); return st; }
    }
    // commit: every replica's two headers around the new tables (the old tables stay allocated until the index is freed).  A
    // publication that fails part-way (replica i) is undone on EVERY replica already switched, replica i's device copies
    // included; if that cannot be done either, the index is marked unusable: shards of one batch must never classify a
    // length differently.
    kmx_status st = KMX_OK;
    struct Saved { KmxIndexDev header; const KmxPlanEntry* fast; uint32_t range; };
    std::vector<Saved> saved;
    for (kmx_index* r : all) saved.push_back({r->h_header, r->d_plan_fast, r->range});
    size_t switched = 0;
    for (size_t i = 0; i < all.size(); ++i) {
        kmx_index* r = all[i];
        hipError_t e = hipSetDevice(r->device);
        r->h_header.plan = d_plans[i];
        r->h_header.range = new_maximum;
        if (d_fast[i]) r->d_plan_fast = d_fast[i];
        r->range = new_maximum;
        switched = i + 1;
        if (e == hipSuccess) e = publish_headers(r);
        if (e != hipSuccess) {
            st = fail(KMX_ERR_HIP, std::string("kmx_index_extend_query_size_range: ") + hipGetErrorString(e));
            break;
        }
    }
    if (st != KMX_OK) {
        const std::string first_err = g_err;
        bool undone = true;
        for (size_t j = 0; j < switched; ++j) {
            kmx_index* r = all[j];
            r->h_header = saved[j].header; r->d_plan_fast = saved[j].fast; r->range = saved[j].range;
            (void)hipGetLastError();
            hipError_t e = hipSetDevice(r->device);
            if (e == hipSuccess) e = publish_headers(r);
            if (e != hipSuccess) undone = false;
        }
        if (!undone) {
            ix->broken = true;
            restore();
            return fail(KMX_ERR_HIP, first_err + "; the previous planner tables could not be restored on every replica: the index is unusable");
        }
        g_err = first_err;
    }
    restore();
    return st;
}

kmx_status kmx_index_arena_host(const kmx_index* ix, const uint32_t** arena, uint64_t* n_elems)
{
    if (!ix || !arena) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_arena_host: NULL argument");
    if (ix->host_arena.empty()) return fail(KMX_ERR_INVALID_ARGUMENT, "index was built without keep_host_arena");
    *arena = ix->host_arena.data();
    if (n_elems) *n_elems = ix->host_arena.size();
    return KMX_OK;
}

kmx_status kmx_index_bucket_host(const kmx_index* cix, uint32_t k, const uint8_t* ranks, const uint32_t** positions, uint32_t* count)
{
    if (!cix || !ranks || !positions || !count) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_bucket_host: NULL argument");
    *positions = nullptr; *count = 0;
    kmx_index* ix = const_cast<kmx_index*>(cix);
    if (ix->host_arena.empty()) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_bucket_host: index was built without keep_host_arena");
    size_t e = 0;
    while (e < ix->ks.size() && ix->ks[e] != k) ++e;
    if (e == ix->ks.size() || !ix->host_dirs) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_bucket_host: the index holds no element for k = " + std::to_string(k));
    uint64_t h = 0;                                              // rank-hash, kmer_index.hpp:56-73 (Horner: no wrap for a valid k)
    for (uint32_t i = 0; i < k; ++i) {
        if (ranks[i] >= ix->sigma) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_bucket_host: a letter outside the alphabet");
        h = h * ix->sigma + ranks[i];
    }
    const KmxElemDev& el = ix->h_header.elems[e];
    kmx_index::HostDir& hd = ix->host_dirs[e];
    std::call_once(hd.once, [&] {
        // one download per element and index: the table the device probes, as the host's directory
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        hipError_t he = hipSetDevice(ix->device);
        try {
            hd.offs.resize(ix->elem_sizes[e].n_offs);
            if (el.table_kind == KMX_TABLE_OPEN) hd.ukeys.resize(ix->elem_sizes[e].n_ukeys);
        } catch (const std::bad_alloc&) {
            hd.st = KMX_ERR_OUT_OF_MEMORY; hd.err = "kmx_index_bucket_host: host allocation failed";
        }
        if (hd.st == KMX_OK) {
            if (he == hipSuccess && !hd.offs.empty()) he = hipMemcpy(hd.offs.data(), el.offs, hd.offs.size() * 4, hipMemcpyDeviceToHost);
            if (he == hipSuccess && !hd.ukeys.empty()) he = hipMemcpy(hd.ukeys.data(), el.ukeys, hd.ukeys.size() * 8, hipMemcpyDeviceToHost);
            if (he != hipSuccess) { hd.st = KMX_ERR_HIP; hd.err = std::string("kmx_index_bucket_host: table download: ") + hipGetErrorString(he); }
        }
        if (have) (void)hipSetDevice(cur);
    });
    if (hd.st != KMX_OK) return fail(hd.st, hd.err);
    uint64_t g = h;                                              // index of the key's group in offs
    if (el.table_kind == KMX_TABLE_OPEN) {
        const auto it = std::lower_bound(hd.ukeys.begin(), hd.ukeys.end(), h);
        if (it == hd.ukeys.end() || *it != h) return KMX_OK;     // at(hash) == nullptr, kmer_index.hpp:76-84
        g = uint64_t(it - hd.ukeys.begin());
    }
    if (g + 1 >= hd.offs.size()) return KMX_OK;
    const uint32_t a = hd.offs[g], b = hd.offs[g + 1];
    if (b == a) return KMX_OK;
    *positions = ix->host_arena.data() + el.arena_base + a;      // the contiguous copy of the groups
    *count = b - a;
    return KMX_OK;
}

kmx_status kmx_index_levels(const kmx_index* ix, uint32_t* levels)
{
    if (!ix || !levels) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_levels: NULL argument");
    for (size_t i = 0; i < ix->ks.size(); ++i) levels[i] = ix->h_header.elems[i].n_levels;
    return KMX_OK;
}

// KMX_CHECKED builds: copies the 16 violation-record words (word 0 = count).
kmx_status kmx_debug_words(const kmx_index* ix, uint64_t* words16)
{
    if (!ix || !words16 || !ix->d_dbg) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_debug_words: NULL argument");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(words16, ix->d_dbg, 16 * 8, hipMemcpyDeviceToHost));
    return KMX_OK;
}

kmx_status kmx_stats_enable(kmx_index* ix, int enable)
{
    if (!ix) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_stats_enable: index is NULL");
    for (kmx_index* peer : ix->peers) (void)kmx_stats_enable(peer, enable);
    std::lock_guard<std::mutex> lock(ix->stats.mu);
    ix->stats.enabled = enable != 0;
    return KMX_OK;
}

kmx_status kmx_stats_get(kmx_index* ix, kmx_kernel_stat* stats, uint32_t* n)
{
    if (!ix || !stats || !n) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_stats_get: NULL argument");
    {
        std::lock_guard<std::mutex> lock(ix->stats.mu);
        (void)hipSetDevice(ix->device);
        ix->stats.drain();
        for (int i = 0; i < K_COUNT; ++i) stats[i] = kmx_kernel_stat{kKernelNames[i], ix->stats.launches[i], ix->stats.ms[i]};
    }
    for (kmx_index* peer : ix->peers) {                       // every replica's launches add up
        std::lock_guard<std::mutex> lock(peer->stats.mu);
        (void)hipSetDevice(peer->device);
        peer->stats.drain();
        for (int i = 0; i < K_COUNT; ++i) { stats[i].launches += peer->stats.launches[i]; stats[i].total_ms += peer->stats.ms[i]; }
    }
    if (!ix->peers.empty()) (void)hipSetDevice(ix->device);
    *n = K_COUNT;
    return KMX_OK;
}

kmx_status kmx_stats_reset(kmx_index* ix)
{
    if (!ix) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_stats_reset: index is NULL");
    for (kmx_index* peer : ix->peers) (void)kmx_stats_reset(peer);
    std::lock_guard<std::mutex> lock(ix->stats.mu);
    (void)hipSetDevice(ix->device);
    ix->stats.drain();
    for (int i = 0; i < K_COUNT; ++i) { ix->stats.launches[i] = 0; ix->stats.ms[i] = 0; }
    return KMX_OK;
}

kmx_status kmx_search_batch_device(const kmx_index* cix, const void* d_qranks, const void* d_qoff, uint64_t nq,
                                   uint32_t flags, void* stream, kmx_result** inout)
{
    if (!cix || !inout) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch_device: NULL argument");
    if (nq && (!d_qranks || !d_qoff)) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch_device: NULL query buffers");
    if (nq >= 0xFFFFFFFFull) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch_device: at most 2^32-2 queries per batch");
    kmx_index* ix = const_cast<kmx_index*>(cix);
    if (ix->broken) return fail(KMX_ERR_HIP, "the index is unusable: a failed kmx_index_extend_query_size_range left its replicas inconsistent");
    if (!ix->peers.empty() && nq) {
        // several replicas: the one that lives where the queries are
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, d_qranks) != hipSuccess) { (void)hipGetLastError(); return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch_device: d_qranks is not a device pointer"); }
        kmx_index* pick = nullptr;
        for (size_t i = 0; i < ix->n_replicas() && !pick; ++i)
            if (ix->replica(i)->device == attr.device) pick = ix->replica(i);
        if (!pick) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch_device: the queries live on a device that holds no replica of this index");
        ix = pick;
    }
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint8_t* qr = static_cast<const uint8_t*>(d_qranks);
    const uint64_t* qo = static_cast<const uint64_t*>(d_qoff);

    kmx_result* r = *inout;
    if (r && !r->parts.empty()) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch_device: the result handle belongs to a multi-device search");
    if (r && r->device != ix->device && r->device_bytes())      // (whatever its last batch was: an empty one still owns buffers there)
        return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch_device: the result handle holds buffers on another device");
    if (!r) { r = take_result(ix); *inout = r; }
    else if (r->ctx.pending) { kmx_status fs = search_finish(r); if (fs != KMX_OK) return fs; }   // its buffers are about to be reused
    r->index = ix;
    r->device = ix->device;
    r->stream = s;
    (void)hipGetLastError();   // do not inherit a stale error from an earlier, unrelated call
    r->flags = flags;
    r->nq = nq;
    r->n_hits = r->n_exact = r->n_stitch = r->n_prefix = r->n_error = r->n_none = r->n_mask_words = 0;
    r->host_valid = r->host_masks_valid = false;
    r->small_valid = false;
    r->quiesced = false;
    if (!r->h_ctr) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&r->h_ctr), KMX_CTR_COUNT * sizeof(unsigned long long), hipHostMallocDefault));
    HIP_TRY(r->hit_off.ensure((nq + 1) * 8));
    if (nq == 0) {
        HIP_TRY(hipMemsetAsync(r->hit_off.p, 0, 8, s));
        return KMX_OK;
    }
    HIP_TRY(r->src.ensure(nq * 8));
    HIP_TRY(r->cnt.ensure(nq * 4));
    HIP_TRY(r->c0.ensure(nq * 4));
    HIP_TRY(r->aux.ensure(nq * 8));
    HIP_TRY(r->key.ensure(nq * 8));
    HIP_TRY(r->p1.ensure(nq * 8));
    HIP_TRY(r->kind.ensure(nq));
    HIP_TRY(r->status.ensure(nq));
    HIP_TRY(r->stitch_list.ensure(nq * 4));
    HIP_TRY(r->prefix_list.ensure(nq * 4));
    HIP_TRY(r->short_list.ensure(nq * 4));
    HIP_TRY(r->bsum.ensure(std::max(kmx::scan_blocks(nq), kmx::lookup_blocks(nq, 4)) * 8));
    if (r->ctr.cap < 2 * KMX_CTR_COUNT * sizeof(unsigned long long)) {
        // two counter blocks per handle, used in turn: the scan of a batch zeroes the block of the next one
        HIP_TRY(r->ctr.ensure(2 * KMX_CTR_COUNT * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(r->ctr.p, 0, 2 * KMX_CTR_COUNT * sizeof(unsigned long long), s));
        r->ctr_phase = 0;
        r->ctr_clean = true;
    }
    kmx::QueryDesc d{r->src.as<uint64_t>(), r->cnt.as<uint32_t>(), r->c0.as<uint32_t>(), r->aux.as<uint64_t>(),
                     r->key.as<uint64_t>(), r->p1.as<uint64_t>(), r->kind.as<uint8_t>(), r->status.as<uint8_t>(),
                     r->stitch_list.as<uint32_t>(), r->prefix_list.as<uint32_t>(), r->short_list.as<uint32_t>(), nullptr};
    // this batch's counter block, and the one of the next batch on this handle
    if (!r->ctr_clean) {                                        // (after a batch whose scan did not reset it: the stream does)
        HIP_TRY(hipMemsetAsync(r->ctr.p, 0, 2 * KMX_CTR_COUNT * sizeof(unsigned long long), s));
        r->ctr_clean = true;
    }
    r->ctr_phase ^= 1u;
    auto* ctr = r->ctr.as<unsigned long long>() + r->ctr_phase * KMX_CTR_COUNT;
    auto* ctr_next = r->ctr.as<unsigned long long>() + (r->ctr_phase ^ 1u) * KMX_CTR_COUNT;
    const KmxIndexDev* dix = header_for(ix, flags);

    // The k_lookup variant, from what the previous batch on this handle held (results do not depend on the choice): batches with
    // two-part cross-referenced queries take the variant that interleaves and finishes them; without, the lean one — with eight
    // queries per thread on an index of tiny cells (a query there is one byte of a table and one 32-byte cell).
    static const int force_items = getenv("KMX_LOOKUP_ITEMS") ? atoi(getenv("KMX_LOOKUP_ITEMS")) : 0;      // (tuning / tests: 4 | 8, or -4: 4 with pairs)
    const bool pairs = force_items ? force_items == -4 : r->last_had_pairs;
    const int items = force_items ? (force_items == 8 ? 8 : 4) : (!pairs && ix->tiny_cells) ? 8 : 4;
    // ... and queries of very many parts (5000-letter reads): when the batch before held some, k_lookup only lists them and
    // k_lookup_long gives each a wave (a lane per part) before the scan reads the counters
    const bool defer_long = r->last_had_long;
    const uint32_t lflags = (flags & ~KMX_SEARCH_INTERNAL_DEFER_LONG) | (defer_long ? KMX_SEARCH_INTERNAL_DEFER_LONG : 0u);
    timed(ix, K_LOOKUP, s, [&] {
        kmx::launch_lookup(s, items, pairs, dix, qr, qo, nq, d, ctr, r->bsum.as<uint64_t>(), lflags);
        if (defer_long) kmx::launch_lookup_long(s, dix, qr, qo, nq, d, ctr, flags);
    });
    // speculative scan: already final when the batch holds no STITCH query
    // The downsweep also records the first query of every output tile (k_partition's job) when the
    // tile table kept from an earlier batch is large enough — the steady state.
    const kmx::FillVariant fv = kmx::effective_fill_variant(ix->fill_variant, ix->rec32);
    const uint64_t tile = kmx::fill_tile(fv);
    const uint64_t tile_cap = r->tile_q.cap / 4;            // entries available while the scan runs
    // the block sums k_lookup left in bsum are the first level of this scan
    // (steady state: the scan's last block also hands the counters to the host and zeroes the next batch's block — no
    //  memset and no copy operation on the stream)
    bool published = false;
    timed(ix, K_SCAN, s, [&] {
        published = kmx::launch_scan_tiles(s, d.cnt, nq, r->bsum.as<uint64_t>(), r->hit_off.as<uint64_t>(), ctr + KMX_CTR_TOTAL_HITS, tile,
                                           tile_cap >= 2 ? tile_cap - 1 : 0, tile_cap >= 2 ? r->tile_q.as<uint32_t>() : nullptr, items,
                                           kmx::CounterPub{ctr, ctr_next, r->h_ctr});
    });
    r->ctr_clean = published;
    // Steady state (tile table and output buffer kept from an earlier batch): k_fill goes out right behind the
    // scan, before the host knows the hit total — it reads the total from device memory and its grid is sized
    // from what the buffers can hold.  Valid whenever the batch holds no STITCH query (their counts come later)
    // and the total fits; both are checked after the read-back, and the regular path below redoes the fill if not.
    // Only attempted when the previous batch on this handle had no STITCH query either.
    const uint64_t spec_tiles = std::min<uint64_t>(tile_cap >= 2 ? tile_cap - 1 : 0, (r->out.cap / 4) / tile);
    const bool spec_fill = !(flags & KMX_SEARCH_COUNT_ONLY) && !r->last_had_stitch && spec_tiles > 0 && spec_tiles < 0x7FFFFFFFull;
    if (spec_fill)
        timed(ix, K_FILL, s, [&] {
            kmx::launch_fill(s, fv, ix->rec32, dix, ix->d_arena, r->hit_off.as<uint64_t>(), r->tile_q.as<uint32_t>(),
                             ctr + KMX_CTR_TOTAL_HITS, spec_tiles, d, r->out.as<uint32_t>());
        });
    if (!published) HIP_TRY(hipMemcpyAsync(r->h_ctr, ctr, KMX_CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    // everything after the read-back lives in search_finish: right away, or (KMX_SEARCH_ASYNC) when the result is next touched
    r->ctx = SearchCtx{ix, qr, qo, s, tile_cap, spec_tiles, spec_fill, true};
    if (flags & KMX_SEARCH_ASYNC) {
        if (!r->done) HIP_TRY(hipEventCreateWithFlags(&r->done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(r->done, s));
        r->pool = ix->pool;                                  // (a handle made for another index keeps its buffers, changes pools)
        std::lock_guard<std::mutex> lock(ix->pool->mu);
        ix->pool->pending.push_back(r);
        return KMX_OK;
    }
    return search_finish(r);
}

// Second half of a search: waits for the counters of the first half, then validates / fills / sorts whatever the
// speculative steady-state path has not already done.  A no-op when nothing is pending.
static kmx_status search_finish(kmx_result* r)
{
    if (!r->ctx.pending) return KMX_OK;
    r->ctx.pending = false;
    kmx_index* ix = r->ctx.ix;
    if (r->flags & KMX_SEARCH_ASYNC) {
        std::lock_guard<std::mutex> lock(ix->pool->mu);
        auto& pend = ix->pool->pending;
        pend.erase(std::remove(pend.begin(), pend.end(), r), pend.end());
    }
    const uint8_t* qr = r->ctx.qr;
    const uint64_t* qo = r->ctx.qo;
    hipStream_t s = r->ctx.s;
    const uint64_t nq = r->nq, tile_cap = r->ctx.tile_cap, spec_tiles = r->ctx.spec_tiles;
    const uint32_t flags = r->flags;
    const bool spec_fill = r->ctx.spec_fill;
    HIP_TRY(hipSetDevice(r->device));
    if (r->done && (flags & KMX_SEARCH_ASYNC)) HIP_TRY(hipEventSynchronize(r->done));
    else HIP_TRY(hipStreamSynchronize(s));
    kmx::QueryDesc d{r->src.as<uint64_t>(), r->cnt.as<uint32_t>(), r->c0.as<uint32_t>(), r->aux.as<uint64_t>(),
                     r->key.as<uint64_t>(), r->p1.as<uint64_t>(), r->kind.as<uint8_t>(), r->status.as<uint8_t>(),
                     r->stitch_list.as<uint32_t>(), r->prefix_list.as<uint32_t>(), r->short_list.as<uint32_t>(), nullptr};
    auto* ctr = r->ctr.as<unsigned long long>() + r->ctr_phase * KMX_CTR_COUNT;     // the counter block this search counts into
    const KmxIndexDev* dix = header_for(ix, flags);
    const kmx::FillVariant fv = kmx::effective_fill_variant(ix->fill_variant, ix->rec32);
    const uint64_t tile = kmx::fill_tile(fv);
    auto scan_hits = [&] {                                     // (k_validate changed the counts: a full scan)
        timed(ix, K_SCAN, s, [&] {
            kmx::launch_scan_tiles(s, d.cnt, nq, r->bsum.as<uint64_t>(), r->hit_off.as<uint64_t>(), ctr + KMX_CTR_TOTAL_HITS, tile,
                                   tile_cap >= 2 ? tile_cap - 1 : 0, tile_cap >= 2 ? r->tile_q.as<uint32_t>() : nullptr, 0);
        });
    };
    const uint64_t n_stitch_groups = r->h_ctr[KMX_CTR_STITCH], n_stitch_tiny = r->h_ctr[KMX_CTR_STITCH_TINY];   // front / back of stitch_list
    const uint64_t n_stitch_short = r->h_ctr[KMX_CTR_STITCH_SHORT];                                     // short_list
    const uint64_t n_stitch_pending = n_stitch_groups + n_stitch_tiny + n_stitch_short;                 // still to be validated
    r->n_stitch = n_stitch_pending + r->h_ctr[KMX_CTR_STITCH_RESOLVED];                                  // (k_lookup resolved the others itself)
    r->last_had_stitch = n_stitch_pending != 0;
    r->last_had_long = r->h_ctr[KMX_CTR_LONG] != 0;
    r->last_had_pairs = r->n_stitch != 0;                        // (resolved ones included: they are what the pairs variant of k_lookup is for)
    const uint64_t n_prefix_small = r->h_ctr[KMX_CTR_PREFIX], n_prefix_big = r->h_ctr[KMX_CTR_PREFIX_BIG];
    r->n_prefix = n_prefix_small + n_prefix_big + r->h_ctr[KMX_CTR_PREFIX_PLAIN];
    r->n_error = r->h_ctr[KMX_CTR_ERROR];
    r->n_none = r->h_ctr[KMX_CTR_NONE];
    r->n_mask_words = r->h_ctr[KMX_CTR_MASK_WORDS];
    r->n_exact = nq - r->n_stitch - r->n_prefix - r->n_error - r->n_none;
    const uint64_t prefix_elems = r->h_ctr[KMX_CTR_PREFIX_ELEMS];
    const uint64_t max_runs = r->h_ctr[KMX_CTR_MAX_RUNS];

    if (n_stitch_pending) {
        HIP_TRY(r->mask_words.ensure(r->n_mask_words * 8));
        // room for every candidate to survive (64 slots per mask word); without it k_compact decodes the masks
        static const bool no_survivors = getenv("KMX_NO_STITCH_HITS") != nullptr;
        const uint64_t n_more = r->h_ctr[KMX_CTR_STITCH_MORE];   // queries whose survivors need their further parts checked
        if ((!(flags & KMX_SEARCH_COUNT_ONLY) || n_more) && !no_survivors && r->stitch_hits.ensure(r->n_mask_words * 64 * 4) == hipSuccess)
            d.stitch_hits = r->stitch_hits.as<uint32_t>();
        else
            (void)hipGetLastError();
        timed(ix, K_VALIDATE, s, [&] {
            kmx::launch_validate(s, dix, ix->d_arena, qr, qo, d, n_stitch_groups, n_more, n_stitch_tiny, d.stitch_list + (nq - n_stitch_tiny),
                                 n_stitch_short, r->mask_words.as<uint64_t>(), !(flags & KMX_SEARCH_KEEP_MASKS));
        });
        scan_hits();
        HIP_TRY(hipMemcpyAsync(r->h_ctr, ctr, KMX_CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    r->n_hits = r->h_ctr[KMX_CTR_TOTAL_HITS];
    if (flags & KMX_SEARCH_COUNT_ONLY) return KMX_OK;

    const uint64_t total = r->n_hits;
    if (total == 0) return KMX_OK;
    const uint64_t n_tiles = (total + tile - 1) / tile;
    if (n_tiles >= 0x7FFFFFFFull) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch_device: result too large, split the batch");
    HIP_TRY(r->out.ensure(total * 4));
    HIP_TRY(r->tile_q.ensure((n_tiles + 1) * 4));
    uint32_t* out = r->out.as<uint32_t>();
    const uint64_t* hit_off = r->hit_off.as<uint64_t>();
    const bool spec_ok = spec_fill && n_stitch_pending == 0 && n_tiles <= spec_tiles;   // the early k_fill already did the work
    if (!spec_ok) {
        if (n_tiles + 1 > tile_cap)   // first batch / the table had to grow: the scan could not fill it
            timed(ix, K_PARTITION, s, [&] { kmx::launch_partition(s, hit_off, nq, tile, n_tiles, r->tile_q.as<uint32_t>()); });
        timed(ix, K_FILL, s, [&] {
            kmx::launch_fill(s, fv, ix->rec32, dix, ix->d_arena, hit_off, r->tile_q.as<uint32_t>(), ctr + KMX_CTR_TOTAL_HITS, n_tiles, d, out);
        });
    }
    if (n_stitch_pending && !d.stitch_hits)
        timed(ix, K_COMPACT, s, [&] {
            if (n_stitch_groups) kmx::launch_compact(s, ix->d_arena, d, n_stitch_groups, r->mask_words.as<uint64_t>(), hit_off, out);
            if (n_stitch_tiny) {
                kmx::QueryDesc dt = d;
                dt.stitch_list = d.stitch_list + (nq - n_stitch_tiny);
                kmx::launch_compact(s, ix->d_arena, dt, n_stitch_tiny, r->mask_words.as<uint64_t>(), hit_off, out);
            }
            if (n_stitch_short) {
                kmx::QueryDesc dt = d;
                dt.stitch_list = d.short_list;
                kmx::launch_compact(s, ix->d_arena, dt, n_stitch_short, r->mask_words.as<uint64_t>(), hit_off, out);
            }
        });

    // PREFIX work list: small queries from the front of prefix_list, the others from its back
    kmx::QueryDesc d_big = d;
    d_big.prefix_list = d.prefix_list + (nq - n_prefix_big);
    const uint64_t n_prefix_merge = r->h_ctr[KMX_CTR_PREFIX_MERGE];       // of the small ones: k_prefix_merge_small's class
    if (n_prefix_small > n_prefix_merge)
        timed(ix, K_PREFIX_SORT_SMALL, s, [&] { kmx::launch_prefix_sort_small(s, dix, qo, d, n_prefix_small, hit_off, ix->d_arena, out); });
    if (n_prefix_merge)
        timed(ix, K_PREFIX_MERGE_SMALL, s, [&] { kmx::launch_prefix_merge_small(s, dix, qo, d, n_prefix_small, hit_off, ix->d_arena, out); });
    if (n_prefix_big) {
        // slices beyond the block kernel's capacity are rows of sorted chunks behind it, merged pairwise in global memory:
        // their tiles are counted first (the chunks of a slice with an odd number of passes start in the scratch buffer)
        const bool large = max_runs > 1 && prefix_elems > 0;
        const uint64_t np = n_prefix_big, T = kmx::prefix_merge_tile();
        const uint64_t max_tiles = large ? prefix_elems / T + np : 0;
        // the slices beyond the 256-thread shape: cut into bands where that works (k_prefix_bands: cut tables + one record per band), the
        // others as chunks (k_prefix_items: one record per chunk — one per slice + one per full chunk at most)
        const uint64_t n_mid = r->h_ctr[KMX_CTR_PREFIX_MID], n_long = np > n_mid ? np - n_mid : 0;
        static const bool no_bands = getenv("KMX_NO_BANDS") != nullptr;                 // (experiments: everything as chunks)
        static const bool no_split = getenv("KMX_NO_SPLIT") != nullptr || no_bands;     // (... no slice spread by value)
        // room for the slices spread by value (k_prefix_split_*): only slices beyond one chunk go there, prefix_elems holds their positions
        const bool split_ok = large && !no_split;
        const uint64_t n_large_max = prefix_elems / KMX_PSORT_BLOCK_CAP + 1;
        kmx::PrefixSplitRoom sr{};
        if (split_ok) {
            sr.cap_splits = n_large_max;
            sr.cap_tiles = prefix_elems / kmx::prefix_split_tile() + n_large_max;
            sr.cap_counters = 3 * (prefix_elems / kmx::prefix_split_target() + n_large_max);
            sr.cap_scratch = prefix_elems + 4 * n_large_max;
        }
        const uint64_t cap_items = n_long ? n_long + prefix_elems / KMX_PSORT_BLOCK_CAP + (split_ok ? prefix_elems / kmx::prefix_split_target() + n_large_max : 0) : 0;
        // (bands are for slices beyond prefix_band_min() positions — one chunk in the build as it ships: a batch without such slices
        //  needs no room for bands and no k_prefix_bands launch)
        const bool chunks_banded = kmx::prefix_band_min() < KMX_PSORT_BLOCK_CAP;
        const bool bands_possible = !no_bands && n_long && (large || chunks_banded);
        const uint64_t cap_bands = !bands_possible ? 0
                                   : chunks_banded ? n_long * (KMX_PSORT_BLOCK_CAP / kmx::prefix_band_target() + 1) + prefix_elems / kmx::prefix_band_target()
                                                   : prefix_elems / kmx::prefix_band_target() + n_large_max;
        const uint64_t cap_cuts = std::min<uint64_t>((cap_bands + (chunks_banded ? n_long : n_large_max)) * kmx::prefix_band_runs(), uint64_t(1) << 26);
        if (n_long) {
            HIP_TRY(r->pitems.ensure(cap_items * kmx::prefix_item_bytes()));
            HIP_TRY(r->pbands.ensure(cap_bands * kmx::prefix_item_bytes()));
            HIP_TRY(r->pcuts.ensure(cap_cuts * 4));
            HIP_TRY(r->pbanded.ensure(np * 4));
            if (split_ok) {
                if (r->psplits.ensure(sr.cap_splits * kmx::prefix_split_bytes(0)) == hipSuccess && r->ptiles.ensure(sr.cap_tiles * kmx::prefix_split_bytes(1)) == hipSuccess &&
                    r->pscnt.ensure(sr.cap_counters * 4) == hipSuccess && r->pscratch.ensure(sr.cap_scratch * 4) == hipSuccess) {
                    sr.splits = r->psplits.p; sr.tiles = r->ptiles.p; sr.counters = r->pscnt.as<uint32_t>(); sr.scratch = r->pscratch.as<uint32_t>();
                    HIP_TRY(hipMemsetAsync(sr.splits, 0, sr.cap_splits * kmx::prefix_split_bytes(0), s));
                    HIP_TRY(hipMemsetAsync(sr.tiles, 0xFF, sr.cap_tiles * kmx::prefix_split_bytes(1), s));
                    HIP_TRY(hipMemsetAsync(sr.counters, 0, sr.cap_counters * 4, s));
                } else {
                    (void)hipGetLastError();                         // no room: those slices go through chunks + merge passes
                    sr = kmx::PrefixSplitRoom{};
                }
            }
            if (!bands_possible && !sr.splits) HIP_TRY(hipMemsetAsync(r->pbanded.p, 0, np * 4, s));
            else {
                timed(ix, K_PREFIX_BANDS, s, [&] {
                    kmx::launch_prefix_bands(s, dix, qo, d_big, np, hit_off, ix->d_arena, r->pbanded.as<uint32_t>(), r->pbands.p, cap_bands,
                                             r->pcuts.as<uint32_t>(), cap_cuts, ctr + KMX_CTR_PSB_BANDS, sr);
                });
                if (sr.splits)
                    timed(ix, K_PREFIX_SPLIT, s, [&] {
                        kmx::launch_prefix_split(s, sr, ctr + KMX_CTR_PSB_BANDS, ix->d_arena, r->pbanded.as<uint32_t>(), ix->h_header.n, r->pitems.p, cap_items,
                                                 ctr + KMX_CTR_PSB_OTHER);
                    });
            }
        }
        if (large) {
            HIP_TRY(r->plen.ensure(np * 4));
            HIP_TRY(r->poff.ensure((np + 1) * 8));
            HIP_TRY(r->ptmp.ensure(max_tiles * T * 4));
            HIP_TRY(r->bsum.ensure(std::max(kmx::scan_blocks(np), kmx::scan_blocks(nq)) * 8));
            timed(ix, K_PREFIX_LEN, s, [&] { kmx::launch_prefix_len(s, d_big, np, r->pbanded.as<uint32_t>(), r->plen.as<uint32_t>()); });
            timed(ix, K_SCAN, s, [&] {
                kmx::launch_scan(s, r->plen.as<uint32_t>(), np, r->bsum.as<uint64_t>(), r->poff.as<uint64_t>(), ctr + KMX_CTR_PREFIX_TOTAL);
            });
        }
        if (n_mid) HIP_TRY(r->pmid.ensure(np * kmx::prefix_item_bytes()));
        timed(ix, K_PREFIX_SORT_BLOCK, s, [&] {
            kmx::launch_prefix_sort_block(s, dix, qo, d_big, np, n_mid, hit_off, ix->d_arena, out, large ? r->poff.as<uint64_t>() : nullptr,
                                          large ? r->ptmp.as<uint32_t>() : nullptr, r->pitems.p, cap_items, ctr + KMX_CTR_PSB_MERGE,
                                          r->pbanded.as<uint32_t>(), r->pbands.p, cap_bands, r->pcuts.as<uint32_t>(), ctr + KMX_CTR_PSB_BANDS, sr.scratch, r->pmid.p, ix->h_header.n, ix->d_dbg);
        });
        if (large) {
            uint32_t passes = 0;
            while ((uint64_t(1) << passes) < max_runs) ++passes;
            for (uint32_t p = 0; p < passes; ++p)
                timed(ix, K_MERGE_PASS, s, [&] {
                    kmx::launch_prefix_merge_pass(s, d_big, np, r->poff.as<uint64_t>(), max_tiles, hit_off, out, r->ptmp.as<uint32_t>(), p);
                });
        }
    }
    HIP_TRY(hipGetLastError());
    return KMX_OK;
}

// host-buffer search of one replica: inputs to the device on the result's own stream, the device form behind them.
// `wait`: return with the search complete; otherwise its second half stays pending (multi-device: every replica is
// started before any is waited for).
static kmx_status search_host_one(kmx_index* ix, const uint8_t* qranks, const uint64_t* qoff, uint64_t q0, uint64_t q1, uint32_t flags,
                                  kmx_result** inout, bool wait)
{
    HIP_TRY(hipSetDevice(ix->device));
    kmx_result* r = *inout ? *inout : take_result(ix);
    *inout = r;
    if (r->ctx.pending) { kmx_status fs = search_finish(r); if (fs != KMX_OK) return fs; }
    if (!r->own_stream) HIP_TRY(hipStreamCreateWithFlags(&r->own_stream, hipStreamNonBlocking));
    const uint64_t nq = q1 - q0;
    const uint64_t l0 = nq ? qoff[q0] : 0, n_letters = nq ? qoff[q1] - l0 : 0;
    // A handful of queries (kmer_index::search(query) is a batch of one; up to 8192 queries in 32 workgroups): one launch that
    // reads the queries from and writes the whole result to page-locked blocks — no copies, no counter read-back, one wait.
    static const bool no_small = getenv("KMX_NO_SMALL") != nullptr;
    const uint32_t n_sb = uint32_t((nq + KMX_SMALL_NQ - 1) / KMX_SMALL_NQ);
    bool small = nq && n_sb <= KMX_SMALL_BLOCKS && !(flags & KMX_SEARCH_COUNT_ONLY) && !no_small;
    KmxSmallArgs sargs{};
    for (uint32_t b = 0; small && b < n_sb; ++b) {
        const uint64_t b0 = q0 + uint64_t(b) * KMX_SMALL_NQ, b1 = std::min(q1, b0 + KMX_SMALL_NQ);
        const uint64_t letters = qoff[b1] - qoff[b0];
        small = (b1 - b0 + 1) * 8 + letters <= KMX_SMALL_IN_BYTES;
        sargs.nq[b] = uint16_t(b1 - b0);
        sargs.n_letters[b] = uint16_t(letters);
        // queries whose length is none of the index's ks are cross-referenced or sub-k ones: more of them than a workgroup
        // takes would only cost a declined launch
        uint32_t maybe_slow = 0;
        for (uint64_t i = b0; small && b1 - b0 > KMX_SMALL_WSLOW && i < b1; ++i)
            maybe_slow += std::find(ix->ks.begin(), ix->ks.end(), uint32_t(qoff[i + 1] - qoff[i])) == ix->ks.end();
        small = small && maybe_slow <= KMX_SMALL_WSLOW + KMX_SMALL_BSLOW;
    }
    // the mailbox is laid out for 1, 4 or 32 workgroups (7 MB page-locked for the largest: allocated once per result handle)
    const KmxSmallLayout L = kmx_small_layout(n_sb <= 1 ? 1u : n_sb <= 4 ? 4u : uint32_t(KMX_SMALL_BLOCKS));
    if (small && n_sb > 1 && r->small_xchg.ensure(KMX_SMALL_BLOCKS * 8) != hipSuccess) { (void)hipGetLastError(); small = false; }
    if (small && r->mailbox.ensure_pinned(L.bytes)) {
        unsigned char* mb0 = r->mailbox.as<unsigned char>();
        for (uint32_t b = 0; b < n_sb; ++b) {
            unsigned char* in = mb0 + size_t(b) * KMX_SMALL_IN_BYTES;
            const uint64_t b0 = q0 + uint64_t(b) * KMX_SMALL_NQ, lb = qoff[b0];
            uint64_t* in_off = reinterpret_cast<uint64_t*>(in);
            for (uint32_t i = 0; i <= sargs.nq[b]; ++i) in_off[i] = qoff[b0 + i] - lb;
            if (sargs.n_letters[b]) memcpy(in + (size_t(sargs.nq[b]) + 1) * 8, qranks + lb, sargs.n_letters[b]);
            reinterpret_cast<KmxSmallHeader*>(mb0 + L.off_header)[b].fallback = 2;          // (overwritten by the kernel)
        }
        (void)hipGetLastError();
        unsigned long long* xchg = n_sb > 1 ? r->small_xchg.as<unsigned long long>() : nullptr;
        if (xchg) HIP_TRY(hipMemsetAsync(xchg, 0, KMX_SMALL_BLOCKS * 8, r->own_stream));
        timed(ix, K_SMALL, r->own_stream, [&] { kmx::launch_small(r->own_stream, header_for(ix, flags), ix->d_arena, mb0, L, n_sb, sargs, uint32_t(nq), xchg, flags); });
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(r->own_stream));
        const KmxSmallHeader* hdrs = reinterpret_cast<const KmxSmallHeader*>(mb0 + L.off_header);
        bool all_ok = true;
        for (uint32_t b = 0; b < n_sb; ++b) all_ok = all_ok && hdrs[b].fallback == 0;
        if (all_ok) {
            r->index = ix; r->device = ix->device; r->stream = r->own_stream; r->flags = flags & ~KMX_SEARCH_ASYNC;
            r->nq = nq;
            r->n_hits = r->n_mask_words = r->n_stitch = r->n_prefix = r->n_error = r->n_none = 0;
            for (uint32_t b = 0; b < n_sb; ++b) {
                r->n_hits += hdrs[b].n_hits; r->n_mask_words += hdrs[b].n_mask_words;
                r->n_stitch += hdrs[b].n_stitch; r->n_prefix += hdrs[b].n_prefix; r->n_error += hdrs[b].n_error; r->n_none += hdrs[b].n_none;
            }
            r->n_exact = nq - r->n_stitch - r->n_prefix - r->n_error - r->n_none;
            // the workgroups wrote into ONE set of arrays: the views are the mailbox
            r->v_hit_off = reinterpret_cast<uint64_t*>(mb0 + L.off_hitoff);
            r->v_positions = reinterpret_cast<uint32_t*>(mb0 + L.off_pos);
            r->v_status = mb0 + L.off_status;
            r->v_kinds = mb0 + L.off_kinds;
            r->m_base = reinterpret_cast<const uint64_t*>(mb0 + L.off_mbase);
            r->m_words = reinterpret_cast<const uint64_t*>(mb0 + L.off_words);
            r->m_ccnt = reinterpret_cast<const uint32_t*>(mb0 + L.off_ccnt);
            r->m_csrc = reinterpret_cast<const uint64_t*>(mb0 + L.off_csrc);
            // the queries themselves, should device views be asked for later (kmx_result_view_device runs the device form then)
            if (!r->small_in.ensure_pageable((nq + 1) * 8 + n_letters + 16)) return fail(KMX_ERR_OUT_OF_MEMORY, "kmx_search_batch: host allocation failed");
            {
                uint64_t* io = r->small_in.as<uint64_t>();
                for (uint64_t i = 0; i <= nq; ++i) io[i] = qoff[q0 + i] - l0;
                if (n_letters) memcpy(r->small_in.as<unsigned char>() + (nq + 1) * 8, qranks + l0, n_letters);
            }
            r->host_valid = r->host_masks_valid = true;
            r->small_valid = true;
            r->quiesced = true;
            r->last_had_stitch = false;
            r->last_had_pairs = false;
            r->last_had_long = false;
            r->pool = ix->pool;                                 // (kmx_result_view_device asks it whether the index is still there)
            return KMX_OK;
        }
        // not a batch for the small kernel (too many hits, long candidate lists): the general path below
    }
    HIP_TRY(r->in_qranks.ensure(std::max<uint64_t>(n_letters, 1) + 16));
    HIP_TRY(r->in_qoff.ensure((nq + 1) * 8));
    if (n_letters) HIP_TRY(hipMemcpyAsync(r->in_qranks.p, qranks + l0, n_letters, hipMemcpyHostToDevice, r->own_stream));
    if (nq) {
        if (l0 == 0) {
            HIP_TRY(hipMemcpyAsync(r->in_qoff.p, qoff + q0, (nq + 1) * 8, hipMemcpyHostToDevice, r->own_stream));
        } else {
            // a shard's offsets are rebased to its own letters
            std::vector<uint64_t> local(nq + 1);
            for (uint64_t i = 0; i <= nq; ++i) local[i] = qoff[q0 + i] - l0;
            HIP_TRY(hipMemcpyAsync(r->in_qoff.p, local.data(), (nq + 1) * 8, hipMemcpyHostToDevice, r->own_stream));
            HIP_TRY(hipStreamSynchronize(r->own_stream));           // `local` goes out of scope
        }
    }
    const uint32_t f = wait ? (flags & ~KMX_SEARCH_ASYNC) : (flags | KMX_SEARCH_ASYNC);
    kmx_status st = kmx_search_batch_device(ix, r->in_qranks.p, r->in_qoff.p, nq, f, r->own_stream, inout);
    if (st != KMX_OK) return st;
    if (wait) {
        HIP_TRY(hipStreamSynchronize(r->own_stream));
        r->quiesced = true;
    }
    return KMX_OK;
}

// A batch too large for one pass over the device (hard part 1: "chunked batches so output fits"): the queries go
// through ONE worker result chunk by chunk; every chunk's result is copied to host memory of its own and the device buffers
// serve the next chunk.  A chunk that still runs out of device memory is halved and tried again.  The parent presents the
// chunks as one result through the merged views of multi-part results; device views do not exist for it.
static kmx_status search_host_chunked(kmx_index* ix, const uint8_t* qranks, const uint64_t* qoff, uint64_t nq, uint32_t flags,
                                      kmx_result** out, uint64_t chunk_q)
{
    kmx_result* parent = *out;
    if (parent && !parent->chunked) {                        // a plain handle from an earlier call: its buffers become the worker's
        kmx_result* fresh = new kmx_result();
        fresh->worker = parent;
        parent = fresh;
    }
    if (!parent) parent = new kmx_result();
    *out = parent;
    parent->chunked = true;
    for (kmx_result* p : parent->parts) kmx_result_free(p);
    parent->parts.clear();
    parent->part_q0.assign(1, 0);
    parent->index = ix; parent->device = ix->device; parent->flags = flags & ~KMX_SEARCH_ASYNC; parent->nq = nq;
    parent->host_valid = parent->host_masks_valid = false;
    parent->n_hits = parent->n_exact = parent->n_stitch = parent->n_prefix = parent->n_error = parent->n_none = parent->n_mask_words = 0;
    const bool masks = (flags & KMX_SEARCH_KEEP_MASKS) != 0;
    chunk_q = std::max<uint64_t>(chunk_q, 1);

    // The PARENT owns the host views of the whole batch (page-locked, grow-only, kept between calls): offsets, statuses and
    // kinds have known sizes; the positions buffer is sized from the first chunk's hits per query (+ 1/8) and grown if a later
    // chunk proves that short.  Chunk i's arrays go device-to-host STRAIGHT into their slices of those views (no staging copy,
    // no per-chunk host memory) on a helper thread, while the calling thread searches chunk i + 1 with the OTHER worker: the
    // PCIe transfer — the long pole of a host-buffer search, 55 GB/s against terabytes per second of search — overlaps the
    // search instead of following it.  A worker is searched into again only after its copy task has been joined.
    const bool want_pos = !(flags & KMX_SEARCH_COUNT_ONLY);
    const uint64_t nq1 = std::max<uint64_t>(nq, 1);
    if (!parent->h_hit_off.ensure((nq + 1) * 8) || !parent->h_status.ensure(nq1) || !parent->h_kinds.ensure(nq1))
        return fail(KMX_ERR_OUT_OF_MEMORY, "kmx_search_batch: host allocation failed");
    parent->v_hit_off = parent->h_hit_off.as<uint64_t>();
    parent->v_status = parent->h_status.as<uint8_t>();
    parent->v_kinds = parent->h_kinds.as<uint8_t>();
    parent->v_positions = parent->h_positions.as<uint32_t>();
    parent->v_hit_off[0] = 0;
    struct CopyTask {
        std::thread th;
        kmx_status st = KMX_OK;
        std::string err;
        bool running = false;
    } tasks[2];
    auto join = [&](int w) -> kmx_status {
        if (tasks[w].running) { tasks[w].th.join(); tasks[w].running = false; }
        if (tasks[w].st != KMX_OK) { g_err = tasks[w].err; const kmx_status st = tasks[w].st; tasks[w].st = KMX_OK; return st; }
        return KMX_OK;
    };
    auto finish = [&](kmx_status st) -> kmx_status {         // no task outlives the call
        for (int w = 0; w < 2; ++w) { const kmx_status js = join(w); if (st == KMX_OK) st = js; }
        return st;
    };
    auto copy_out = [&](int wi, kmx_result* w, kmx_result* part, uint64_t q0, uint64_t cq, uint64_t h0) {
        CopyTask& t = tasks[wi];
        hipError_t e = hipSetDevice(ix->device);
        hipStream_t cs = w->stream;
        // (search_host_one left the worker complete: counters read, every kernel of the chunk done)
        if (e == hipSuccess && cq) e = hipMemcpyAsync(parent->v_hit_off + q0 + 1, w->hit_off.as<uint64_t>() + 1, cq * 8, hipMemcpyDeviceToHost, cs);
        if (e == hipSuccess && cq) e = hipMemcpyAsync(parent->v_status + q0, w->status.p, cq, hipMemcpyDeviceToHost, cs);
        if (e == hipSuccess && cq) e = hipMemcpyAsync(parent->v_kinds + q0, w->kind.p, cq, hipMemcpyDeviceToHost, cs);
        if (e == hipSuccess && want_pos && w->n_hits) e = hipMemcpyAsync(parent->v_positions + h0, w->out.p, w->n_hits * 4, hipMemcpyDeviceToHost, cs);
        if (e == hipSuccess) e = hipStreamSynchronize(cs);
        if (e != hipSuccess) { t.st = KMX_ERR_HIP; t.err = std::string("kmx_search_batch: chunk copy: ") + hipGetErrorString(e); return; }
        if (h0)
            for (uint64_t q = q0 + 1; q <= q0 + cq; ++q) parent->v_hit_off[q] += h0;
        if (masks) {
            const uint64_t* mb = nullptr; const uint64_t* mw = nullptr; const uint32_t* cc = nullptr; const uint64_t* cs2 = nullptr;
            const kmx_status st = kmx_result_masks(w, &mb, &mw, &cc, &cs2);
            if (st != KMX_OK) { t.st = st; t.err = g_err; return; }
            if (!part->h_mask_base.ensure_pageable((cq + 1) * 8) || !part->h_cand_count.ensure_pageable((cq + 1) * 4) ||
                !part->h_cand_src.ensure_pageable((cq + 1) * 8) || !part->h_mask_words.ensure_pageable((w->n_mask_words + 1) * 8)) {
                t.st = KMX_ERR_OUT_OF_MEMORY; t.err = "kmx_search_batch: host allocation for a chunk's masks failed";
                return;
            }
            memcpy(part->h_mask_base.p, mb, cq * 8); memcpy(part->h_cand_count.p, cc, cq * 4); memcpy(part->h_cand_src.p, cs2, cq * 8);
            if (w->n_mask_words) memcpy(part->h_mask_words.p, mw, w->n_mask_words * 8);
            part->m_base = part->h_mask_base.as<uint64_t>(); part->m_words = part->h_mask_words.as<uint64_t>();
            part->m_ccnt = part->h_cand_count.as<uint32_t>(); part->m_csrc = part->h_cand_src.as<uint64_t>();
        }
    };
    int turn = 0;
    bool single_worker = false;                               // after an out-of-memory: ONE set of device buffers from there on
    static const long long inject_oom = getenv("KMX_TEST_INJECT_CHUNK_OOM") ? atoll(getenv("KMX_TEST_INJECT_CHUNK_OOM")) : 0;   // (test hook: the n-th chunk search reports out of memory)
    long long n_searches = 0;
    uint64_t h0 = 0;                                          // hits of the chunks in front
    for (uint64_t q0 = 0; q0 < nq;) {
        const uint64_t q1 = std::min(nq, q0 + chunk_q);
        kmx_result*& wr = turn ? parent->worker2 : parent->worker;
        kmx_status st = join(turn);                          // the worker's previous chunk has left it
        if (st != KMX_OK) return finish(st);
        st = (inject_oom && ++n_searches == inject_oom) ? fail(KMX_ERR_OUT_OF_MEMORY, "kmx_search_batch: injected out-of-memory (KMX_TEST_INJECT_CHUNK_OOM)")
                                                        : search_host_one(ix, qranks, qoff, q0, q1, flags, &wr, true);
        if (st == KMX_ERR_OUT_OF_MEMORY && (chunk_q > 1024 || !single_worker)) {
            // Out of device memory.  First the second set of device buffers goes (both copy tasks joined, worker2 released whichever
            // worker was being searched into: its grow-only buffers from the larger chunks would starve the retry) and the SAME chunk
            // size is tried with one worker; only if one worker alone does not fit either are the chunks halved.
            (void)hipGetLastError();
            const kmx_status js = finish(KMX_OK);
            if (js != KMX_OK) return js;
            if (single_worker) chunk_q = std::max<uint64_t>(chunk_q / 2, 1024);
            single_worker = true;
            if (parent->worker2) { kmx_result_free(parent->worker2); parent->worker2 = nullptr; }
            turn = 0;
            {
                // results parked in the pool hold device memory too
                std::vector<kmx_result*> idle;
                { std::lock_guard<std::mutex> lock(ix->pool->mu); idle.swap(ix->pool->idle); }
                for (kmx_result* r : idle) { r->release(); delete r; }
            }
            continue;
        }
        if (st != KMX_OK) return finish(st);
        kmx_result* w = wr;
        if (w->small_valid) {
            // (a chunk small enough for the one-launch latency path leaves nothing in HBM: its views are host memory already)
            const uint64_t* ho; const uint32_t* pos; const uint8_t* stt; const uint8_t* kd;
            if ((st = kmx_result_view(w, &ho, &pos, &stt, &kd)) != KMX_OK) return finish(st);
        }
        const uint64_t cq = q1 - q0;
        if (want_pos && (h0 + w->n_hits) * 4 > parent->h_positions.cap) {
            // the positions view: room for this chunk and, by its hits per query so far, for the rest of the batch
            if ((st = finish(KMX_OK)) != KMX_OK) return st;  // (copies into the old buffer are done)
            const uint64_t sofar = h0 + w->n_hits;
            const uint64_t guess = q1 < nq ? sofar + (sofar / std::max<uint64_t>(q1, 1) + 1) * (nq - q1) * 9 / 8 : sofar;
            HostBuf bigger;
            if (!bigger.ensure(std::max<uint64_t>(guess, 1) * 4)) return fail(KMX_ERR_OUT_OF_MEMORY, "kmx_search_batch: host allocation for the hit lists failed");
            if (h0) memcpy(bigger.p, parent->h_positions.p, h0 * 4);
            parent->h_positions.release();
            parent->h_positions = bigger;
            bigger.p = nullptr; bigger.cap = 0;               // (moved)
            parent->v_positions = parent->h_positions.as<uint32_t>();
        }
        auto* part = new kmx_result();
        parent->parts.push_back(part);
        parent->part_q0.push_back(q1);
        part->host_chunk = true; part->host_valid = true; part->host_masks_valid = masks; part->quiesced = true;
        part->device = ix->device; part->flags = parent->flags; part->nq = cq;
        part->n_hits = w->n_hits; part->n_exact = w->n_exact; part->n_stitch = w->n_stitch; part->n_prefix = w->n_prefix;
        part->n_error = w->n_error; part->n_none = w->n_none; part->n_mask_words = w->n_mask_words;
        parent->n_hits += part->n_hits; parent->n_exact += part->n_exact; parent->n_stitch += part->n_stitch; parent->n_prefix += part->n_prefix;
        parent->n_error += part->n_error; parent->n_none += part->n_none;
        if (w->small_valid) {                                // host to host, here and now
            memcpy(parent->v_hit_off + q0 + 1, w->v_hit_off + 1, cq * 8);
            memcpy(parent->v_status + q0, w->v_status, cq);
            memcpy(parent->v_kinds + q0, w->v_kinds, cq);
            if (want_pos && w->n_hits) memcpy(parent->v_positions + h0, w->v_positions, w->n_hits * 4);
            if (h0)
                for (uint64_t q = q0 + 1; q <= q1; ++q) parent->v_hit_off[q] += h0;
            if (masks) {
                auto masks_only = [&, turn, w, part, q0, cq] {
                    // (masks only: reuse the copy task with nothing else left to move)
                    CopyTask& t = tasks[turn];
                    const uint64_t* mb = nullptr; const uint64_t* mw = nullptr; const uint32_t* cc = nullptr; const uint64_t* cs2 = nullptr;
                    const kmx_status ms = kmx_result_masks(w, &mb, &mw, &cc, &cs2);
                    if (ms != KMX_OK) { t.st = ms; t.err = g_err; return; }
                    if (!part->h_mask_base.ensure_pageable((cq + 1) * 8) || !part->h_cand_count.ensure_pageable((cq + 1) * 4) ||
                        !part->h_cand_src.ensure_pageable((cq + 1) * 8) || !part->h_mask_words.ensure_pageable((w->n_mask_words + 1) * 8)) {
                        t.st = KMX_ERR_OUT_OF_MEMORY; t.err = "kmx_search_batch: host allocation for a chunk's masks failed"; return;
                    }
                    memcpy(part->h_mask_base.p, mb, cq * 8); memcpy(part->h_cand_count.p, cc, cq * 4); memcpy(part->h_cand_src.p, cs2, cq * 8);
                    if (w->n_mask_words) memcpy(part->h_mask_words.p, mw, w->n_mask_words * 8);
                    part->m_base = part->h_mask_base.as<uint64_t>(); part->m_words = part->h_mask_words.as<uint64_t>();
                    part->m_ccnt = part->h_cand_count.as<uint32_t>(); part->m_csrc = part->h_cand_src.as<uint64_t>();
                    (void)q0;
                };
                try {
                    tasks[turn].th = std::thread(masks_only);
                    tasks[turn].running = true;
                } catch (const std::system_error&) {
                    masks_only();
                }
            }
        } else {
            try {
                tasks[turn].th = std::thread(copy_out, turn, w, part, q0, cq, h0);
                tasks[turn].running = true;
            } catch (const std::system_error&) {                // no thread to be had: the copies run here, in line
                copy_out(turn, w, part, q0, cq, h0);
            }
        }
        h0 += w->n_hits;
        q0 = q1;
        if (!single_worker) turn ^= 1;                       // (one worker: its copy task is joined at the top of the loop before it is searched into again)
    }
    const kmx_status fs = finish(KMX_OK);
    if (fs != KMX_OK) return fs;
    parent->host_valid = true;                               // the merged views ARE the chunks' destination
    return KMX_OK;
}

kmx_status kmx_search_batch(const kmx_index* cix, const uint8_t* qranks, const uint64_t* qoff, uint64_t nq,
                            uint32_t flags, kmx_result** out)
{
    if (!cix || !out) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch: NULL argument");
    if (nq && !qoff) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch: NULL query offsets");
    if (nq && qoff[0] != 0) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch: qoff[0] must be 0");
    for (uint64_t i = 0; i < nq; ++i)
        if (qoff[i + 1] < qoff[i]) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch: qoff must be non-decreasing");
    if (nq && !qranks && qoff[nq] != 0) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch: NULL query letters");   // (a batch of empty queries has none)
    kmx_index* ix = const_cast<kmx_index*>(cix);
    if (ix->broken) return fail(KMX_ERR_HIP, "the index is unusable: a failed kmx_index_extend_query_size_range left its replicas inconsistent");
    const size_t W = ix->n_replicas();
    if (W == 1) {
        if (*out && !(*out)->parts.empty() && !(*out)->chunked) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch: the result handle belongs to a multi-device search");
        uint64_t chunk_q = uint64_t(1) << 25;                   // queries per pass; KMX_HOST_CHUNK overrides (tests)
        if (const char* e = getenv("KMX_HOST_CHUNK")) { const long long v = atoll(e); if (v > 0) chunk_q = uint64_t(v); }
        if (nq > chunk_q || (*out && (*out)->chunked)) return search_host_chunked(ix, qranks, qoff, nq, flags, out, chunk_q);
        const bool fresh = *out == nullptr;
        kmx_status st = search_host_one(ix, qranks, qoff, 0, nq, flags, out, true);
        if (st == KMX_ERR_OUT_OF_MEMORY && nq > 4096) {          // the batch does not fit the device in one pass: stream it
            (void)hipGetLastError();
            if (fresh && *out) { kmx_result_free(*out); *out = nullptr; }
            return search_host_chunked(ix, qranks, qoff, nq, flags, out, std::max<uint64_t>(nq / 4, 1024));
        }
        return st;
    }
    // several replicas: replica r searches the contiguous range [nq*r/W, nq*(r+1)/W) on its own device and
    // stream; all of them are started before the first is waited for.  The parent result presents the parts as one.
    int caller_device = 0;
    (void)hipGetDevice(&caller_device);
    kmx_result* parent = *out;
    if (parent && parent->chunked) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch: the result handle belongs to a chunk-streamed search of another index");
    if (parent && parent->parts.size() != W) {
        if (!parent->parts.empty() || parent->device_bytes()) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_search_batch: the result handle was made by a search of another shape");
    }
    if (!parent) { parent = new kmx_result(); *out = parent; }
    parent->index = ix;
    parent->device = ix->device;
    parent->flags = flags & ~KMX_SEARCH_ASYNC;
    parent->nq = nq;
    parent->host_valid = parent->host_masks_valid = false;
    parent->parts.resize(W, nullptr);
    parent->part_q0.assign(W + 1, 0);
    for (size_t r = 0; r <= W; ++r) parent->part_q0[r] = nq / W * r + (nq % W) * r / W;   // == floor(nq * r / W) without the overflow
    if (nq < 256 * W)                                   // a handful of queries: one device, one round trip (the other parts stay empty)
        for (size_t r = 1; r <= W; ++r) parent->part_q0[r] = nq;
    // one host thread per replica for the first half (input copies from pageable memory are staged by the calling thread:
    // in turn they would reach the GPUs one after the other, each device waiting for the one before it)
    kmx_status st = KMX_OK;
    {
        std::vector<kmx_status> sts(W, KMX_OK);
        std::vector<std::string> msgs(W);
        std::vector<std::thread> threads;
        size_t busy = 0;
        for (size_t r = 0; r < W; ++r) busy += parent->part_q0[r + 1] > parent->part_q0[r];
        auto run = [&](size_t r) {
            sts[r] = search_host_one(ix->replica(r), qranks, qoff, parent->part_q0[r], parent->part_q0[r + 1], flags, &parent->parts[r], false);
            if (sts[r] != KMX_OK) msgs[r] = g_err;                       // (g_err is thread-local)
        };
        for (size_t r = 0; r < W; ++r) {
            if (busy > 1 && parent->part_q0[r + 1] > parent->part_q0[r]) threads.emplace_back(run, r);
            else run(r);
        }
        for (auto& t : threads) t.join();
        for (size_t r = 0; r < W && st == KMX_OK; ++r)
            if (sts[r] != KMX_OK) st = fail(sts[r], msgs[r]);
    }
    parent->n_hits = parent->n_exact = parent->n_stitch = parent->n_prefix = parent->n_error = parent->n_none = parent->n_mask_words = 0;
    for (size_t r = 0; r < W; ++r) {
        kmx_result* p = parent->parts[r];
        if (!p) continue;
        const kmx_status fs = search_finish(p);
        if (st == KMX_OK) st = fs;
        parent->n_hits += p->n_hits; parent->n_exact += p->n_exact; parent->n_stitch += p->n_stitch; parent->n_prefix += p->n_prefix;
        parent->n_error += p->n_error; parent->n_none += p->n_none;
    }
    (void)hipSetDevice(caller_device);
    return st;
}

kmx_status kmx_result_counts(const kmx_result* r, uint64_t* nq, uint64_t* n_hits, uint64_t* n_exact, uint64_t* n_stitch,
                             uint64_t* n_prefix, uint64_t* n_error)
{
    if (!r) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_counts: result is NULL");
    if (r->ctx.pending) { kmx_status fs = search_finish(const_cast<kmx_result*>(r)); if (fs != KMX_OK) return fs; }
    if (nq) *nq = r->nq;
    if (n_hits) *n_hits = r->n_hits;
    if (n_exact) *n_exact = r->n_exact;
    if (n_stitch) *n_stitch = r->n_stitch;
    if (n_prefix) *n_prefix = r->n_prefix;
    if (n_error) *n_error = r->n_error;
    return KMX_OK;
}

kmx_status kmx_result_parts(const kmx_result* r, uint32_t* n_parts)
{
    if (!r || !n_parts) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_parts: NULL argument");
    *n_parts = r->parts.empty() ? 1u : uint32_t(r->parts.size());
    return KMX_OK;
}

kmx_status kmx_result_part_view_device(const kmx_result* r, uint32_t part, int32_t* device, uint64_t* q_begin, uint64_t* q_end,
                                       const uint64_t** d_hit_off, const uint32_t** d_positions, const uint8_t** d_status)
{
    if (!r) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_part_view_device: result is NULL");
    const uint32_t n_parts = r->parts.empty() ? 1u : uint32_t(r->parts.size());
    if (part >= n_parts) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_part_view_device: no such part");
    const kmx_result* p = r->parts.empty() ? r : r->parts[part];
    if (device) *device = p->device;
    if (q_begin) *q_begin = r->parts.empty() ? 0 : r->part_q0[part];
    if (q_end) *q_end = r->parts.empty() ? r->nq : r->part_q0[part + 1];
    return kmx_result_view_device(p, d_hit_off, d_positions, d_status);
}

kmx_status kmx_result_view_device(const kmx_result* r, const uint64_t** d_hit_off, const uint32_t** d_positions,
                                  const uint8_t** d_status)
{
    if (!r) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_view_device: result is NULL");
    if (r->chunked) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_view_device: the batch was streamed through the device in chunks: its result lives in host memory only");
    if (!r->parts.empty()) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_view_device: the result spans several devices: use kmx_result_part_view_device");
    if (r->host_chunk) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_view_device: the batch was streamed through the device in chunks: its result lives in host memory only");
    if (r->small_valid) {
        // the last search ran on the latency path and left nothing in HBM: run the same queries (they are still in the
        // mailbox) through the device form now
        kmx_result* rr = const_cast<kmx_result*>(r);
        {
            bool gone = !rr->pool;
            if (rr->pool) { std::lock_guard<std::mutex> lock(rr->pool->mu); gone = rr->pool->closed; }
            if (gone) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_view_device: the result was produced on the latency path (host memory only) and its index has been freed");
        }
        kmx_index* ix = const_cast<kmx_index*>(rr->index);
        const unsigned char* mb = rr->small_in.as<unsigned char>();
        const uint64_t nq = rr->nq, n_letters = reinterpret_cast<const uint64_t*>(mb)[nq];
        HIP_TRY(hipSetDevice(rr->device));
        HIP_TRY(rr->in_qranks.ensure(std::max<uint64_t>(n_letters, 1) + 16));
        HIP_TRY(rr->in_qoff.ensure((nq + 1) * 8));
        HIP_TRY(hipMemcpyAsync(rr->in_qoff.p, mb, (nq + 1) * 8, hipMemcpyHostToDevice, rr->own_stream));
        if (n_letters) HIP_TRY(hipMemcpyAsync(rr->in_qranks.p, mb + (nq + 1) * 8, n_letters, hipMemcpyHostToDevice, rr->own_stream));
        kmx_result* self = rr;
        const kmx_status st = kmx_search_batch_device(ix, rr->in_qranks.p, rr->in_qoff.p, nq, rr->flags, rr->own_stream, &self);
        if (st != KMX_OK) return st;
    }
    if (r->ctx.pending) { kmx_status fs = search_finish(const_cast<kmx_result*>(r)); if (fs != KMX_OK) return fs; }
    if (d_hit_off) *d_hit_off = r->hit_off.as<uint64_t>();
    if (d_positions) *d_positions = r->out.as<uint32_t>();
    if (d_status) *d_status = r->status.as<uint8_t>();
    return KMX_OK;
}

kmx_status kmx_result_gather_device(kmx_result* r, int32_t dst_device, const uint64_t** d_hit_off, const uint32_t** d_positions,
                                    const uint8_t** d_status)
{
    if (!r) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_gather_device: result is NULL");
    if (r->chunked || r->host_chunk)
        return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_gather_device: the batch was streamed through the device in chunks: its result lives in host memory only");
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (dst_device < 0 || dst_device >= n_dev) return fail(KMX_ERR_NO_DEVICE, "kmx_result_gather_device: no such device");
    if (r->ctx.pending) { kmx_status fs = search_finish(r); if (fs != KMX_OK) return fs; }
    if (r->parts.empty() && r->device == dst_device) return kmx_result_view_device(r, d_hit_off, d_positions, d_status);   // already there
    int caller_device = 0;
    (void)hipGetDevice(&caller_device);
    struct Part { const kmx_result* p; uint64_t q0, nq; const uint64_t* ho; const uint32_t* pos; const uint8_t* st; hipEvent_t ev; };
    std::vector<Part> parts;
    auto leave = [&](kmx_status st) {
        for (Part& pt : parts) if (pt.ev) (void)hipEventDestroy(pt.ev);
        (void)hipSetDevice(caller_device);
        return st;
    };
    const size_t n_parts = r->parts.empty() ? 1 : r->parts.size();
    for (size_t i = 0; i < n_parts; ++i) {
        kmx_result* p = r->parts.empty() ? r : r->parts[i];
        const uint64_t q0 = r->parts.empty() ? 0 : r->part_q0[i], q1 = r->parts.empty() ? r->nq : r->part_q0[i + 1];
        if (!p || q1 == q0) continue;
        Part pt{p, q0, q1 - q0, nullptr, nullptr, nullptr, nullptr};
        const kmx_status st = kmx_result_view_device(p, &pt.ho, &pt.pos, &pt.st);     // (completes the part; a latency-path part is re-run on the device)
        if (st != KMX_OK) return leave(st);
        parts.push_back(pt);
    }
    const bool want_pos = !(r->flags & KMX_SEARCH_COUNT_ONLY) && r->n_hits;
    hipError_t e = hipSetDevice(dst_device);
    if (e == hipSuccess && r->gather_device >= 0 && r->gather_device != dst_device) {
        // (buffers of an earlier gather to another device)
        (void)hipSetDevice(r->gather_device);
        r->g_hit_off.release(); r->g_out.release(); r->g_status.release();
        if (r->gather_stream) (void)hipStreamDestroy(r->gather_stream);
        r->gather_stream = nullptr;
        e = hipSetDevice(dst_device);
    }
    r->gather_device = dst_device;
    if (e == hipSuccess && !r->gather_stream) e = hipStreamCreateWithFlags(&r->gather_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = r->g_hit_off.ensure((r->nq + 1) * 8);
    if (e == hipSuccess) e = r->g_status.ensure(std::max<uint64_t>(r->nq, 1));
    if (e == hipSuccess && want_pos) e = r->g_out.ensure(r->n_hits * 4);
    if (e != hipSuccess) return leave(fail(e == hipErrorOutOfMemory ? KMX_ERR_OUT_OF_MEMORY : KMX_ERR_HIP, std::string("kmx_result_gather_device: ") + hipGetErrorString(e)));
    uint64_t* g_off = r->g_hit_off.as<uint64_t>();
    // every part leaves on its OWN stream (behind its last kernel) and over its own link ...
    uint64_t h0 = 0;
    std::vector<uint64_t> base;
    for (Part& pt : parts) {
        const kmx_result* p = pt.p;
        base.push_back(h0);
        e = hipSetDevice(p->device);
        if (e == hipSuccess) e = hipMemcpyPeerAsync(g_off + pt.q0 + 1, dst_device, pt.ho + 1, p->device, pt.nq * 8, p->stream);
        if (e == hipSuccess) e = hipMemcpyPeerAsync(r->g_status.as<uint8_t>() + pt.q0, dst_device, pt.st, p->device, pt.nq, p->stream);
        if (e == hipSuccess && want_pos && p->n_hits)
            e = hipMemcpyPeerAsync(r->g_out.as<uint32_t>() + h0, dst_device, pt.pos, p->device, p->n_hits * 4, p->stream);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&pt.ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(pt.ev, p->stream);
        if (e != hipSuccess) return leave(fail(KMX_ERR_HIP, std::string("kmx_result_gather_device: part copy: ") + hipGetErrorString(e)));
        h0 += p->n_hits;
    }
    // ... and the destination rebases each part's offsets behind the hits of the parts in front of it
    e = hipSetDevice(dst_device);
    if (e == hipSuccess) e = hipMemsetAsync(g_off, 0, 8, r->gather_stream);
    for (size_t i = 0; i < parts.size() && e == hipSuccess; ++i) {
        e = hipStreamWaitEvent(r->gather_stream, parts[i].ev, 0);
        if (e == hipSuccess) {
            kmx::launch_rebase_offsets(r->gather_stream, g_off + parts[i].q0 + 1, parts[i].nq, base[i]);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(r->gather_stream);
    if (e != hipSuccess) return leave(fail(KMX_ERR_HIP, std::string("kmx_result_gather_device: ") + hipGetErrorString(e)));
    if (d_hit_off) *d_hit_off = g_off;
    if (d_positions) *d_positions = want_pos ? r->g_out.as<uint32_t>() : nullptr;
    if (d_status) *d_status = r->g_status.as<uint8_t>();
    return leave(KMX_OK);
}

kmx_status kmx_result_view(kmx_result* r, const uint64_t** hit_off, const uint32_t** positions, const uint8_t** status,
                           const uint8_t** kinds)
{
    if (!r) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_view: result is NULL");
    if (r->ctx.pending) { kmx_status fs = search_finish(r); if (fs != KMX_OK) return fs; }
    if ((!r->parts.empty() || r->chunked) && !r->host_valid) {
        // the parts of a multi-device result, concatenated in replica order: every device copies straight into its slice
        // of the one host buffer (all links at once), the offsets are rebased on the host
        const bool have_pos = !(r->flags & KMX_SEARCH_COUNT_ONLY) && r->n_hits;
        // (a chunk-streamed batch never gets here: its chunks were copied into the parent's views as they were searched)
        if (!r->h_hit_off.ensure((r->nq + 1) * 8) || !r->h_status.ensure(std::max<uint64_t>(r->nq, 1)) ||
            !r->h_kinds.ensure(std::max<uint64_t>(r->nq, 1)) || !r->h_positions.ensure(std::max<uint64_t>(have_pos ? r->n_hits * 4 : 0, 4)))
            return fail(KMX_ERR_OUT_OF_MEMORY, "kmx_result_view: host allocation failed");
        r->v_hit_off = r->h_hit_off.as<uint64_t>();
        r->v_positions = r->h_positions.as<uint32_t>();
        r->v_status = r->h_status.as<uint8_t>();
        r->v_kinds = r->h_kinds.as<uint8_t>();
        int caller_device = 0;
        (void)hipGetDevice(&caller_device);
        uint64_t h0 = 0;
        r->v_hit_off[0] = 0;
        for (size_t i = 0; i < r->parts.size(); ++i) {
            kmx_result* p = r->parts[i];
            const uint64_t q0 = r->part_q0[i], nqp = r->part_q0[i + 1] - q0;
            HIP_TRY(hipSetDevice(p->device));
            if (p->small_valid) {                                   // the part's result already is on the host
                memcpy(r->v_hit_off + q0 + 1, p->v_hit_off + 1, nqp * 8);
                memcpy(r->v_status + q0, p->v_status, nqp);
                memcpy(r->v_kinds + q0, p->v_kinds, nqp);
                if (have_pos && p->n_hits) memcpy(r->v_positions + h0, p->v_positions, p->n_hits * 4);
                h0 += p->n_hits;
                continue;
            }
            if (nqp) {
                HIP_TRY(hipMemcpyAsync(r->v_hit_off + q0 + 1, p->hit_off.as<uint64_t>() + 1, nqp * 8, hipMemcpyDeviceToHost, p->stream));
                HIP_TRY(hipMemcpyAsync(r->v_status + q0, p->status.p, nqp, hipMemcpyDeviceToHost, p->stream));
                HIP_TRY(hipMemcpyAsync(r->v_kinds + q0, p->kind.p, nqp, hipMemcpyDeviceToHost, p->stream));
            }
            if (have_pos && p->n_hits) HIP_TRY(hipMemcpyAsync(r->v_positions + h0, p->out.p, p->n_hits * 4, hipMemcpyDeviceToHost, p->stream));
            h0 += p->n_hits;
        }
        h0 = 0;
        for (size_t i = 0; i < r->parts.size(); ++i) {
            kmx_result* p = r->parts[i];
            HIP_TRY(hipSetDevice(p->device));
            if (!p->small_valid && r->part_q0[i + 1] > r->part_q0[i]) HIP_TRY(hipStreamSynchronize(p->stream));
            p->quiesced = true;
            if (h0)
                for (uint64_t q = r->part_q0[i] + 1; q <= r->part_q0[i + 1]; ++q) r->v_hit_off[q] += h0;
            h0 += p->n_hits;
        }
        (void)hipSetDevice(caller_device);
        r->host_valid = true;
    }
    if (!r->host_valid) {
        HIP_TRY(hipSetDevice(r->device));
        const bool have_pos = !(r->flags & KMX_SEARCH_COUNT_ONLY) && r->n_hits;
        const size_t b_off = (r->nq + 1) * 8, b_pos = have_pos ? r->n_hits * 4 : 0, b_st = r->nq;
        const size_t small_total = b_off + ((b_pos + 7) & ~size_t(7)) + 2 * ((b_st + 7) & ~size_t(7));
        if (small_total <= kSmallView && r->h_small.ensure_pinned(kSmallView)) {
            // small result: the four arrays share one page-locked block, four async copies, one wait
            char* base = r->h_small.as<char>();
            r->v_hit_off = reinterpret_cast<uint64_t*>(base);
            r->v_positions = reinterpret_cast<uint32_t*>(base + b_off);
            r->v_status = reinterpret_cast<uint8_t*>(base + b_off + ((b_pos + 7) & ~size_t(7)));
            r->v_kinds = r->v_status + ((b_st + 7) & ~size_t(7));
            HIP_TRY(hipMemcpyAsync(r->v_hit_off, r->hit_off.p, b_off, hipMemcpyDeviceToHost, r->stream));
            if (r->nq) {
                HIP_TRY(hipMemcpyAsync(r->v_status, r->status.p, r->nq, hipMemcpyDeviceToHost, r->stream));
                HIP_TRY(hipMemcpyAsync(r->v_kinds, r->kind.p, r->nq, hipMemcpyDeviceToHost, r->stream));
            }
            if (have_pos) HIP_TRY(hipMemcpyAsync(r->v_positions, r->out.p, b_pos, hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(hipStreamSynchronize(r->stream));
        } else {
            HIP_TRY(hipStreamSynchronize(r->stream));
            if (!r->h_hit_off.ensure(b_off) || !r->h_status.ensure(std::max<uint64_t>(r->nq, 1)) ||
                !r->h_kinds.ensure(std::max<uint64_t>(r->nq, 1)) || !r->h_positions.ensure(std::max<uint64_t>(b_pos, 4)))
                return fail(KMX_ERR_OUT_OF_MEMORY, "kmx_result_view: host allocation failed");
            r->v_hit_off = r->h_hit_off.as<uint64_t>();
            r->v_positions = r->h_positions.as<uint32_t>();
            r->v_status = r->h_status.as<uint8_t>();
            r->v_kinds = r->h_kinds.as<uint8_t>();
            HIP_TRY(hipMemcpy(r->v_hit_off, r->hit_off.p, b_off, hipMemcpyDeviceToHost));
            if (r->nq) {
                HIP_TRY(hipMemcpy(r->v_status, r->status.p, r->nq, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(r->v_kinds, r->kind.p, r->nq, hipMemcpyDeviceToHost));
            }
            if (have_pos) HIP_TRY(hipMemcpy(r->v_positions, r->out.p, b_pos, hipMemcpyDeviceToHost));
        }
        r->quiesced = true;
        r->host_valid = true;
    }
    if (hit_off) *hit_off = r->v_hit_off;
    if (positions) *positions = ((r->flags & KMX_SEARCH_COUNT_ONLY) || !r->n_hits) ? nullptr : r->v_positions;
    if (status) *status = r->v_status;
    if (kinds) *kinds = r->v_kinds;
    return KMX_OK;
}

kmx_status kmx_result_masks(kmx_result* r, const uint64_t** mask_base, const uint64_t** mask_words,
                            const uint32_t** cand_count, const uint64_t** cand_src)
{
    if (!r) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_masks: result is NULL");
    if (r->ctx.pending) { kmx_status fs = search_finish(r); if (fs != KMX_OK) return fs; }
    if (!(r->flags & KMX_SEARCH_KEEP_MASKS)) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_result_masks: search ran without KMX_SEARCH_KEEP_MASKS");
    if ((!r->parts.empty() || r->chunked) && !r->host_masks_valid) {
        // per-part mask views, concatenated; a part's mask_base counts from its own first word, cand_src is an arena
        // index and the same in every replica
        uint64_t n_words = 0;
        r->part_w0.assign(r->parts.size() + 1, 0);
        for (size_t i = 0; i < r->parts.size(); ++i) { n_words += r->parts[i]->n_mask_words; r->part_w0[i + 1] = n_words; }
        r->n_mask_words = n_words;
        const uint64_t nq1 = std::max<uint64_t>(r->nq, 1);
        if (!r->h_mask_base.ensure(nq1 * 8) || !r->h_cand_count.ensure(nq1 * 4) || !r->h_cand_src.ensure(nq1 * 8) ||
            !r->h_mask_words.ensure(std::max<uint64_t>(n_words, 1) * 8))
            return fail(KMX_ERR_OUT_OF_MEMORY, "kmx_result_masks: host allocation failed");
        int caller_device = 0;
        (void)hipGetDevice(&caller_device);
        for (size_t i = 0; i < r->parts.size(); ++i) {
            const uint64_t* mb; const uint64_t* mw; const uint32_t* cc; const uint64_t* cs;
            const kmx_status st = kmx_result_masks(r->parts[i], &mb, &mw, &cc, &cs);
            if (st != KMX_OK) { (void)hipSetDevice(caller_device); return st; }
            const uint64_t q0 = r->part_q0[i], nqp = r->part_q0[i + 1] - q0, w0 = r->part_w0[i];
            for (uint64_t q = 0; q < nqp; ++q) r->h_mask_base.as<uint64_t>()[q0 + q] = mb[q] + w0;
            if (nqp) { memcpy(r->h_cand_count.as<uint32_t>() + q0, cc, nqp * 4); memcpy(r->h_cand_src.as<uint64_t>() + q0, cs, nqp * 8); }
            if (r->parts[i]->n_mask_words) memcpy(r->h_mask_words.as<uint64_t>() + w0, mw, r->parts[i]->n_mask_words * 8);
        }
        (void)hipSetDevice(caller_device);
        r->host_masks_valid = true;
    }
    if (!r->host_masks_valid) {
        HIP_TRY(hipSetDevice(r->device));
        HIP_TRY(hipStreamSynchronize(r->stream));
        const uint64_t nq1 = std::max<uint64_t>(r->nq, 1);
        if (!r->h_mask_base.ensure(nq1 * 8) || !r->h_cand_count.ensure(nq1 * 4) || !r->h_cand_src.ensure(nq1 * 8) ||
            !r->h_mask_words.ensure(std::max<uint64_t>(r->n_mask_words, 1) * 8))
            return fail(KMX_ERR_OUT_OF_MEMORY, "kmx_result_masks: host allocation failed");
        if (r->nq) {
            HIP_TRY(hipMemcpy(r->h_mask_base.p, r->aux.p, r->nq * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(r->h_cand_count.p, r->c0.p, r->nq * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(r->h_cand_src.p, r->src.p, r->nq * 8, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < r->nq; ++i) r->h_cand_src.as<uint64_t>()[i] &= ~(uint64_t(3) << 62);   // internal flags
        }
        if (r->n_mask_words) HIP_TRY(hipMemcpy(r->h_mask_words.p, r->mask_words.p, r->n_mask_words * 8, hipMemcpyDeviceToHost));
        r->host_masks_valid = true;
    }
    if (!r->small_valid) {
        r->m_base = r->h_mask_base.as<uint64_t>(); r->m_words = r->h_mask_words.as<uint64_t>();
        r->m_ccnt = r->h_cand_count.as<uint32_t>(); r->m_csrc = r->h_cand_src.as<uint64_t>();
    }
    if (mask_base) *mask_base = r->m_base;
    if (mask_words) *mask_words = r->m_words;
    if (cand_count) *cand_count = r->m_ccnt;
    if (cand_src) *cand_src = r->m_csrc;
    return KMX_OK;
}

void kmx_result_free(kmx_result* r)
{
    if (!r) return;
    if (!r->parts.empty() || r->chunked) {
        for (kmx_result* p : r->parts) kmx_result_free(p);     // each part returns to its replica's pool
        r->parts.clear();
        if (r->worker) kmx_result_free(r->worker);
        if (r->worker2) kmx_result_free(r->worker2);
        r->worker = r->worker2 = nullptr;
        r->release();
        delete r;
        return;
    }
    if (r->ctx.pending) (void)search_finish(r);
    if (r->pool && r->device_bytes() <= kPoolMaxBytes) {
        if (!r->quiesced) {                                  // its buffers are about to serve another stream
            (void)hipSetDevice(r->device);
            (void)hipDeviceSynchronize();
            r->quiesced = true;
        }
        std::lock_guard<std::mutex> lock(r->pool->mu);
        if (!r->pool->closed && r->pool->idle.size() < kPoolMaxResults) {
            r->pool->idle.push_back(r);
            return;
        }
    }
    if (r->index) (void)hipSetDevice(r->device);
    r->release();
    delete r;
}

} // extern "C"

// ---------------------------------------------------------------------------
// On-disk image of the flattened index (the thesis states the intent — "the index is serialized so
// it can be loaded directly at a later point", thesis/content/02_implementation.tex:44-46 — the
// reference never implemented it).  Layout, all little endian, sections padded to 8 bytes:
//   FileHeader | FileElem[n_ks] | tail[kmax] | per element: positions, offs, slots, ukeys
// The checksum covers everything after the header.
// ---------------------------------------------------------------------------
namespace {

struct FileHeader {
    char magic[8];          // "KMXIMG01"
    uint32_t version;       // 2
    uint32_t sigma;
    uint64_t n;
    uint32_t n_ks;
    uint32_t range;
    uint32_t kmax;
    uint32_t reserved;
    uint64_t checksum;
};
struct FileElem {
    uint32_t k, table_kind, log2cap, reserved;
    uint64_t n_keys, npos, n_offs, n_slots, n_ukeys, region, n_aoffs;
};

struct Mixer {                       // word-wise 64-bit checksum (order sensitive)
    uint64_t h = 0x9E3779B97F4A7C15ull;
    void add(const void* p, size_t bytes)
    {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        size_t i = 0;
        for (; i + 8 <= bytes; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x100000001B3ull; h ^= h >> 29; }
        uint64_t w = 0;
        if (i < bytes) { memcpy(&w, b + i, bytes - i); h = (h ^ w) * 0x100000001B3ull; h ^= h >> 29; }
    }
};

bool write_section(FILE* f, Mixer& mx, const void* p, size_t bytes)
{
    static const char zero[8] = {0};
    mx.add(p, bytes);
    if (bytes && fwrite(p, 1, bytes, f) != bytes) return false;
    size_t pad = (8 - bytes % 8) % 8;
    return pad == 0 || fwrite(zero, 1, pad, f) == pad;
}

bool read_section(FILE* f, Mixer& mx, void* p, size_t bytes)
{
    char skip[8];
    if (bytes && fread(p, 1, bytes, f) != bytes) return false;
    mx.add(p, bytes);
    size_t pad = (8 - bytes % 8) % 8;
    return pad == 0 || fread(skip, 1, pad, f) == pad;
}

} // namespace

extern "C" kmx_status kmx_index_save(const kmx_index* ix, const char* path)
{
    if (!ix || !path) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_save: NULL argument");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());
    FILE* f = fopen(path, "wb");
    if (!f) return fail(KMX_ERR_INVALID_ARGUMENT, std::string("kmx_index_save: cannot open ") + path);
    const uint32_t n_ks = uint32_t(ix->ks.size());
    FileHeader fh{};
    memcpy(fh.magic, "KMXIMG01", 8);
    fh.version = 2; fh.sigma = ix->sigma; fh.n = ix->n; fh.n_ks = n_ks; fh.range = ix->range; fh.kmax = ix->h_header.kmax;
    bool ok = fwrite(&fh, sizeof fh, 1, f) == 1;
    Mixer mx;
    std::vector<FileElem> fes(n_ks);
    for (uint32_t i = 0; i < n_ks; ++i) {
        const KmxElemDev& el = ix->h_header.elems[i];
        fes[i] = FileElem{el.k, el.table_kind, el.log2cap, 0, el.n_keys, el.npos, ix->elem_sizes[i].n_offs, ix->elem_sizes[i].n_slots,
                          ix->elem_sizes[i].n_ukeys, el.region, ix->elem_sizes[i].n_aoffs};
    }
    ok = ok && write_section(f, mx, fes.data(), fes.size() * sizeof(FileElem));
    ok = ok && write_section(f, mx, ix->tail.data(), ix->tail.size());
    std::vector<unsigned char> buf;
    auto dump = [&](const void* dptr, size_t bytes) -> kmx_status {
        buf.resize(bytes);
        if (bytes) HIP_TRY(hipMemcpy(buf.data(), dptr, bytes, hipMemcpyDeviceToHost));
        if (!write_section(f, mx, buf.data(), bytes)) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_save: write failed");
        return KMX_OK;
    };
    kmx_status st = ok ? KMX_OK : fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_save: write failed");
    for (uint32_t i = 0; i < n_ks && st == KMX_OK; ++i) {
        const KmxElemDev& el = ix->h_header.elems[i];
        st = dump(ix->d_arena + el.arena_base, el.region * 4);
        if (st == KMX_OK) st = dump(el.offs, ix->elem_sizes[i].n_offs * 4);
        if (st == KMX_OK) st = dump(el.atab, ix->elem_sizes[i].n_aoffs * 4);
        if (st == KMX_OK) st = dump(el.slots, ix->elem_sizes[i].n_slots * sizeof(KmxSlot));
        if (st == KMX_OK) st = dump(el.ukeys, ix->elem_sizes[i].n_ukeys * 8);
    }
    if (st == KMX_OK) {
        fh.checksum = mx.h;
        if (fseek(f, 0, SEEK_SET) != 0 || fwrite(&fh, sizeof fh, 1, f) != 1) st = fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_save: write failed");
    }
    if (fclose(f) != 0 && st == KMX_OK) st = fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_save: close failed");
    return st;
}

extern "C" kmx_status kmx_index_load(const char* path, const kmx_options* opts, kmx_index** out)
{
    if (!path || !out) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: NULL argument");
    *out = nullptr;
    kmx_options o;
    if (!read_options(opts, o)) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: options.struct_size mismatch");
    if (o.n_devices > 1) o.device = o.devices[0];
    FILE* f = fopen(path, "rb");
    if (!f) return fail(KMX_ERR_INVALID_ARGUMENT, std::string("kmx_index_load: cannot open ") + path);
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    FileHeader fh{};
    if (fread(&fh, sizeof fh, 1, f) != 1 || memcmp(fh.magic, "KMXIMG01", 8) != 0 || fh.version != 2)
        return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: not a kmx index image (bad magic / version)");
    if (fh.n_ks == 0 || fh.n_ks > KMX_MAX_KS || fh.kmax == 0 || fh.kmax > 63 || fh.n < fh.kmax || fh.n + fh.kmax - 1 >= 0xFFFFFFFFull ||
        fh.range == 0 || fh.range > 65535 * 9u)
        return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: corrupt header");
    Mixer mx;
    std::vector<FileElem> fes(fh.n_ks);
    if (!read_section(f, mx, fes.data(), fes.size() * sizeof(FileElem))) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: truncated file");
    uint32_t kmax = 0;
    for (const FileElem& fe : fes) {
        const bool dense = fe.table_kind == KMX_TABLE_DENSE, open = fe.table_kind == KMX_TABLE_OPEN;
        if (!kmx::k_is_valid(fh.sigma, fe.k) || fe.npos != fh.n - fe.k + 1 || fe.n_keys != kmx::key_space(fh.sigma, fe.k) || !(dense || open) ||
            (dense && (fe.n_offs != fe.n_keys + 1 || fe.n_slots || fe.n_ukeys)) ||
            (open && (fe.n_offs != fe.n_ukeys + 1 || fe.log2cap > 40 || fe.n_slots != (uint64_t(1) << fe.log2cap) || fe.n_ukeys > fe.npos)) ||
            fe.region < fe.npos || fe.region >= 0xFFFFFFFFull || (fe.n_aoffs != 0 && !(dense && fe.n_aoffs == fe.n_keys + 1 && fe.region > fe.npos)))
            return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: corrupt element table");
        kmax = std::max(kmax, fe.k);
    }
    if (kmax != fh.kmax) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: corrupt header (kmax)");
    {
        // The element table promises sizes; nothing is allocated on its word before the file is seen to hold exactly
        // that much (a 60-byte file may not ask for 16 GB of positions or a 2^40-slot table).
        const long here = ftell(f);
        if (here < 0 || fseek(f, 0, SEEK_END) != 0) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: cannot size the file");
        const long end = ftell(f);
        if (end < here || fseek(f, here, SEEK_SET) != 0) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: cannot size the file");
        const uint64_t left = uint64_t(end - here);
        auto pad8 = [](unsigned __int128 b) { return (b + 7) / 8 * 8; };
        unsigned __int128 want = pad8(fh.kmax);
        for (const FileElem& fe : fes)
            want += pad8((unsigned __int128)fe.region * 4) + pad8((unsigned __int128)fe.n_offs * 4) + pad8((unsigned __int128)fe.n_aoffs * 4) +
                    pad8((unsigned __int128)fe.n_slots * sizeof(KmxSlot)) + pad8((unsigned __int128)fe.n_ukeys * 8);
        if (want != left)
            return fail(KMX_ERR_INVALID_ARGUMENT, want > left ? "kmx_index_load: truncated file" : "kmx_index_load: the file is longer than its element table says");
    }
    std::vector<uint8_t> tail(fh.kmax);
    if (!read_section(f, mx, tail.data(), tail.size())) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: truncated file");
    std::vector<kmx::ElemImage> images(fh.n_ks);
    for (uint32_t i = 0; i < fh.n_ks; ++i) {
        const FileElem& fe = fes[i];
        kmx::ElemImage& im = images[i];
        im.k = fe.k; im.table_kind = fe.table_kind; im.log2cap = fe.log2cap; im.n_keys = fe.n_keys; im.npos = fe.npos; im.region = fe.region;
        try {
            im.positions.resize(fe.region); im.offs.resize(fe.n_offs); im.slots.resize(fe.n_slots); im.ukeys.resize(fe.n_ukeys);
            im.atab.resize(fe.n_aoffs);
        } catch (const std::bad_alloc&) {
            return fail(KMX_ERR_OUT_OF_MEMORY, "kmx_index_load: host allocation failed");
        }
        if (!read_section(f, mx, im.positions.data(), fe.region * 4) || !read_section(f, mx, im.offs.data(), fe.n_offs * 4) ||
            !read_section(f, mx, im.atab.data(), fe.n_aoffs * 4) ||
            !read_section(f, mx, im.slots.data(), fe.n_slots * sizeof(KmxSlot)) || !read_section(f, mx, im.ukeys.data(), fe.n_ukeys * 8))
            return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: truncated file");
    }
    if (mx.h != fh.checksum) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: checksum mismatch (corrupt image)");
    for (uint8_t c : tail)
        if (c >= fh.sigma) return fail(KMX_ERR_INVALID_ARGUMENT, "kmx_index_load: corrupt contents: a letter of the text tail is outside the alphabet");
    // The checksum is not cryptographic and says nothing about an image that was WRITTEN wrong: check what the kernels
    // index with.  A full slot table would make probe() spin for ever, an offset past the region reads out of bounds.
    for (uint32_t i = 0; i < fh.n_ks; ++i) {
        const kmx::ElemImage& im = images[i];
        auto bad = [&](const char* what) { return fail(KMX_ERR_INVALID_ARGUMENT, std::string("kmx_index_load: corrupt contents (k = ") + std::to_string(im.k) + "): " + what); };
        if (im.offs.empty() || im.offs.front() != 0 || im.offs.back() != im.npos) return bad("group boundaries do not span the positions");
        for (size_t j = 1; j < im.offs.size(); ++j)
            if (im.offs[j] < im.offs[j - 1]) return bad("group boundaries are not monotone");
        const uint64_t last_start = fh.n - im.k;
        for (uint64_t j = 0; j < im.npos; ++j)
            if (im.positions[j] > last_start) return bad("a position lies outside the text");
        for (size_t j = 0; j + 1 < im.offs.size(); ++j)                  // the kernels search and merge groups as sorted runs
            for (uint64_t t = uint64_t(im.offs[j]) + 1; t < im.offs[j + 1]; ++t)
                if (im.positions[t] <= im.positions[t - 1]) return bad("the positions of a group are not strictly ascending");
        if (im.table_kind == KMX_TABLE_OPEN) {
            const uint64_t cap = uint64_t(1) << im.log2cap;
            if (2 * uint64_t(im.ukeys.size()) > cap) return bad("the open-addressing table is more than half full");
            for (size_t j = 0; j < im.ukeys.size(); ++j) {
                if (im.ukeys[j] >= im.n_keys || (j && im.ukeys[j] <= im.ukeys[j - 1])) return bad("distinct keys are not strictly ascending inside the key space");
                if (im.offs[j + 1] == im.offs[j]) return bad("a listed key owns no position");
            }
            uint64_t used = 0;
            for (const KmxSlot& sl : im.slots) {
                if (!sl.cnt) continue;
                ++used;
                if (sl.key >= im.n_keys || uint64_t(sl.off) + sl.cnt > im.region) return bad("a slot points outside the element's region");
                const size_t g = size_t(std::lower_bound(im.ukeys.begin(), im.ukeys.end(), sl.key) - im.ukeys.begin());
                if (g == im.ukeys.size() || im.ukeys[g] != sl.key || sl.off != im.offs[g] || sl.cnt != im.offs[g + 1] - im.offs[g])
                    return bad("a slot does not name the group of its key");
            }
            if (used != im.ukeys.size()) return bad("occupied slots and distinct keys differ in number");
        } else if (!im.atab.empty()) {
            for (size_t j = 0; j < im.atab.size(); ++j) {
                if ((im.atab[j] & ~31u) > im.region) return bad("an aligned-table entry points outside the element's region");
                if (j && (im.atab[j] & ~31u) < (im.atab[j - 1] & ~31u)) return bad("the aligned table is not monotone");
            }
            if ((im.atab.front() & ~31u) < im.npos) return bad("the aligned copy overlaps the contiguous copy");
            for (size_t j = 0; j + 1 < im.atab.size(); ++j) {             // the aligned copy restates the groups, nothing else
                const uint32_t e0 = im.atab[j], e1 = im.atab[j + 1];
                const uint32_t padded = (e1 & ~31u) - (e0 & ~31u), r = e0 & 31u;
                const uint32_t c = padded ? (r ? padded - 32u + r : padded) : 0u;      // atab_count() of the kernels
                if (c != im.offs[j + 1] - im.offs[j] || uint64_t(e0 & ~31u) + c > im.region ||
                    (c && memcmp(&im.positions[e0 & ~31u], &im.positions[im.offs[j]], size_t(c) * 4) != 0))
                    return bad("the aligned copy of a group differs from the group");
            }
        }
    }
    kmx_status st = check_device();
    if (st != KMX_OK) return st;
    int device = o.device;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    HIP_TRY(hipSetDevice(device));
    st = install_images_impl(images, tail.data(), fh.n, fh.sigma, o.query_size_range ? o.query_size_range : fh.range, device, o, out, nullptr);
    if (st != KMX_OK) return st;
    st = add_prefix_levels(*out, o);
    if (st == KMX_OK) st = add_replicas(*out, o);
    if (st != KMX_OK) { std::string keep = g_err; kmx_index_free(*out); *out = nullptr; g_err = keep; }
    return st;
}
