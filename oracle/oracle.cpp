// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's k-mer exact-match search path
// (Clemapfel/kmer_index), written from a reading of the reference sources.
// Every function cites the reference file:line it follows (paths relative to
// /root/reference).  Nothing under kmer_index_amd/ (the product) may include,
// link or call this file: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, and only as the checker / reported CPU baseline.
//
// PINNING STATUS
//   * orc_fast_pow, orc_bitset_*      : pinned against the real reference headers
//                                        compiled from where they lie (oracle/_ref,
//                                        see ref_shim.cpp) and against
//                                        tests/golden/fast_pow.json, bitset.json.
//   * orc_plan (choose_search_scheme) : pinned against the thesis' known-answer
//                                        table (thesis/content/03_measuring_performance.tex:109-128).
//   * orc_search* (kmer_index::search): PARITY UNPINNED against reference outputs:
//                                        kmer_index.hpp cannot be built here (it needs
//                                        seqan3 and robin_hood.h, neither vendored nor
//                                        installed) and the reference ships no golden
//                                        vectors.  It is anchored instead on the
//                                        reference's own test contract
//                                        (test_main.cpp:37-45: search(q).to_vector()
//                                        == the exact occurrence list) through
//                                        orc_naive_scan, inside the envelope where the
//                                        reference is correct.
//
// Two search modes:
//   ORC_MODE_FAITHFUL : restates the reference's control flow line by line,
//                       including its >=3-part defects (kmer_index.hpp:314,
//                       :526, :535) with the two undefined dereferences
//                       (:317, :546) given the defined meaning "mismatch",
//                       and the 32-bit `int last_hash` of :214-226 (a false
//                       "same hash as the previous part" once sigma^k > 2^32).
//   ORC_MODE_INTENDED : the same algorithm with those lines repaired
//                       (what to_vector() is documented to mean: every text
//                       offset where the query occurs, ascending).
// Inside the SURVEY §4.3 envelope both modes and the naive scan agree.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <thread>
#include <unordered_map>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------
// fast_pow — fast_pow.hpp:10-44 (bit-length LUT) and :46-93 (fall-through
// square-and-multiply).  The LUT maps exp -> number of significant bits for
// exp < 63 and 255 ("overflow") for exp >= 63; note entry 63 is already 255
// (fast_pow.hpp:19), so 2^63 evaluates to 0.
// ---------------------------------------------------------------------------
static inline uint8_t highest_bit_set(uint8_t exp)
{
    if (exp >= 63) return 255;            // fast_pow.hpp:19-43
    uint8_t bits = 0;
    while (exp) { ++bits; exp >>= 1; }    // fast_pow.hpp:12-19 (0,1,2,2,3,3,3,3,4...)
    return bits;
}

static inline uint64_t fast_pow(uint64_t base, uint8_t exp)
{
    uint64_t result = 1;
    uint8_t steps = highest_bit_set(exp);
    if (steps == 255)                     // fast_pow.hpp:54-60
        return base == 1 ? 1 : 0;
    // fast_pow.hpp:62-91: `steps` rounds of (multiply if low bit; shift; square),
    // the last round without the trailing shift/square.
    for (uint8_t s = steps; s >= 1; --s) {
        if (exp & 1) result *= base;
        if (s > 1) { exp >>= 1; base *= base; }
    }
    return result;
}

// ---------------------------------------------------------------------------
// compressed_bitset<uint_fast64_t> — compressed_bitset.hpp:9-105.
// n_bits/64 + 1 words (:23), every word filled with all-ones or zero (:23),
// bit i lives in word i>>6 at bit i&63 (:13-14, :49-50, :59-60, :69-70);
// set_0/set_1/at throw std::out_of_range past n_bits (:46, :56, :66).
// ---------------------------------------------------------------------------
struct bitset {
    uint64_t n_bits;
    std::vector<uint64_t> words;
    bitset(uint64_t n, bool ones)
        : n_bits(n), words(std::max<uint64_t>(n / 64 + 1, 1), ones ? ~uint64_t(0) : 0) {}
    void set_0(uint64_t i) { check(i); words[i >> 6] &= ~(uint64_t(1) << (i & 63)); }
    void set_1(uint64_t i) { check(i); words[i >> 6] |= uint64_t(1) << (i & 63); }
    bool at(uint64_t i) const { check(i); return (words[i >> 6] >> (i & 63)) & 1; }
    uint64_t count(bool b) const        // compressed_bitset.hpp:94-104 (per-bit loop)
    {
        uint64_t ones = 0;
        for (uint64_t i = 0; i < n_bits; ++i) ones += at(i);
        return b ? ones : n_bits - ones;
    }
private:
    void check(uint64_t i) const
    {
        if (i >= n_bits) throw std::out_of_range("compressed bitset index out of range");
    }
};

using bucket_t = std::vector<uint32_t>;

// ---------------------------------------------------------------------------
// kmer_index_result<uint32_t> — kmer_index_result.hpp:15-272.
// ---------------------------------------------------------------------------
struct result {
    bitset mask;                              // _bitmask            :18
    bool bypass;                              // _bypass_bitmask     :19
    uint64_t n_results;                       // _n_results          :20
    std::vector<const bucket_t*> positions;   // _positions          :23

    result() : mask(0, true), bypass(false), n_results(0) {}                       // :203-206
    result(const bucket_t* pos, bool fill, bool bypass_)                           // :208-212
        : mask(bypass_ ? 0 : pos->size(), fill), bypass(bypass_), n_results(pos->size())
    { positions = {pos}; }
    explicit result(std::vector<const bucket_t*> pos)                              // :214-225
        : mask(0, true), bypass(true), n_results(0)
    {
        for (const auto* v : pos) n_results += v->size();
        positions = std::move(pos);
    }
    void should_not_use(uint64_t i) { mask.set_0(i); }                             // :228-231
    void should_use(uint64_t i) { mask.set_1(i); }                                 // :233-236
    bool is_valid(uint64_t i) const { return bypass ? true : mask.at(i); }         // :188-194
    uint64_t size() const { return mask.count(true); }                             // :239-242
    std::vector<uint32_t> to_vector() const                                        // :244-260
    {
        std::vector<uint32_t> out;
        if (positions.empty()) return out;
        uint64_t i = 0;
        for (const auto* vec : positions)
            for (uint64_t j = 0; j < vec->size(); ++j, ++i)
                if (is_valid(i)) out.push_back((*vec)[j]);
        std::sort(out.begin(), out.end());
        return out;
    }
};

enum { MODE_FAITHFUL = 0, MODE_INTENDED = 1 };
enum { ST_OK = 0, ST_TOO_LONG = 1, ST_FANOUT = 2, ST_EMPTY_QUERY = 3 };

struct search_error { int status; };

// ---------------------------------------------------------------------------
// kmer_index_element<alphabet_t, uint32_t, k> — kmer_index.hpp:39-347, with
// sigma and k as runtime members instead of template constants.
// ---------------------------------------------------------------------------
struct element {
    uint32_t sigma = 0, k = 0;
    std::unordered_map<uint64_t, bucket_t> data;      // _data :52 (robin_hood there)
    std::vector<uint8_t> last_kmer;                   // _last_kmer :87
    std::vector<bucket_t> last_kmer_refs;             // _last_kmer_refs :88

    // hash — :56-73: sum over i<k of rank(q_i) * fast_pow(sigma, k-i-1).
    uint64_t hash(const uint8_t* q) const
    {
        uint64_t h = 0;
        for (uint32_t i = 0; i < k; ++i) h += uint64_t(q[i]) * fast_pow(sigma, uint8_t(k - i - 1));
        return h;
    }
    // at — :76-84
    const bucket_t* at(uint64_t h) const
    {
        auto it = data.find(h);
        return it != data.end() ? &it->second : nullptr;
    }
    // check_last_kmer — :90-112: offsets i in [1, k-size] of the last k-mer
    // where the sub-k query matches add the one-element bucket {n-k+i}.
    void check_last_kmer(const uint8_t* q, uint64_t size, std::vector<const bucket_t*>& fill) const
    {
        for (uint64_t i = 1; i < k - size + 1; ++i) {
            bool equal = true;
            for (uint64_t j = i; j < i + size; ++j)
                if (last_kmer.at(j) != q[j - i]) { equal = false; break; }
            if (equal) fill.push_back(&last_kmer_refs.at(i));
        }
    }
    // get_position_for_all_kmer_with_prefix — :115-148
    std::vector<const bucket_t*> prefix_buckets(const uint8_t* q, uint64_t size) const
    {
        if (double(fast_pow(sigma, uint8_t(k - size))) > 1e7) throw search_error{ST_FANOUT};   // :119-122
        uint64_t prefix_hash = 0;
        for (uint64_t i = 0; i < size; ++i)
            prefix_hash += uint64_t(q[i]) * fast_pow(sigma, uint8_t(k - i - 1));               // :126-129
        uint64_t lo = prefix_hash, n_hashes = fast_pow(sigma, uint8_t(k - size));
        std::vector<const bucket_t*> out;
        for (uint64_t h = lo; h < lo + n_hashes; ++h) {                                        // :138-144
            const auto* pos = at(h);
            if (pos) out.push_back(pos);
        }
        check_last_kmer(q, size, out);                                                         // :146
        return out;
    }
    // create — :154-179 (seqan3::views::kmer_hash :157 yields the same polynomial
    // as hash(); thesis/content/02_implementation.tex:10-19).
    void create(const uint8_t* text, uint64_t n)
    {
        uint64_t i = 0;
        if (n >= k) {
            uint64_t top = fast_pow(sigma, uint8_t(k - 1));
            uint64_t h = hash(text);
            for (;;) {
                data[h].push_back(uint32_t(i));                                                // :160-167
                ++i;
                if (i + k > n) break;
                h = (h - uint64_t(text[i - 1]) * top) * sigma + text[i + k - 1];
            }
        }
        uint32_t text_size = uint32_t(i) + k - 1;                                              // :172
        last_kmer.assign(text + n - k, text + n);                                              // :174
        last_kmer_refs.clear();
        for (uint32_t j = 0; j < last_kmer.size(); ++j)
            last_kmer_refs.push_back(bucket_t{j + text_size - k});                             // :177-178
    }
    // search_k — :183-190
    const bucket_t* search_k(const uint8_t* q) const { return at(hash(q)); }

    // search — :193-346
    result search(const uint8_t* q, uint64_t m, int mode) const
    {
        if (m == k) {                                                                          // :198-205
            const auto* pos = at(hash(q));
            return pos ? result(pos, true, true) : result();
        }
        if (m > k) {                                                                           // :207-339
            uint64_t rest_n = m % k;
            std::vector<const bucket_t*> nk;
            // :214 `int last_hash = -1;` — :219 compares the size_t hash with that int (the int is converted to size_t,
            // i.e. SIGN-EXTENDED), :226 stores the hash back into it (truncated to 32 bits).  Consequences, restated exactly in
            // FAITHFUL mode: a previous hash in [2^31, 2^32) never compares equal (shortcut off, harmless); a previous hash
            // >= 2^32 whose bit 31 is clear compares equal to the hash `prev & 0xFFFFFFFF` — a DIFFERENT k-mer — and the
            // reference then reuses the PREVIOUS part's bucket for it (defect 4; reachable once sigma^k > 2^32, e.g. DNA4 k >= 17).
            // INTENDED mode keeps the full hash: the shortcut only ever skips a lookup that would return the same bucket.
            int32_t last_hash = -1;
            bool have_last = false;
            uint64_t last_full = 0;
            for (uint64_t i = 0; i < m - rest_n; i += k) {                                     // :216-227
                uint64_t h = hash(q + i);
                const bool same = (mode == MODE_FAITHFUL) ? (h == uint64_t(int64_t(last_hash)) && !nk.empty())
                                                          : (have_last && h == last_full);
                const auto* pos = same ? nk.back() : at(h);                                    // :219
                if (!pos) return result();
                nk.push_back(pos);
                last_hash = int32_t(uint32_t(h));                                              // :226
                last_full = h; have_last = true;
            }
            bitset usable(nk.back()->size(), true);                                            // :230
            if (rest_n > 0) {                                                                  // :232-256
                auto rest = prefix_buckets(q + m - rest_n, rest_n);
                uint64_t i = 0;
                for (uint32_t pos : *nk.back()) {
                    bool ok = false;
                    for (const auto* rv : rest)
                        if (std::binary_search(rv->begin(), rv->end(), pos + k)) { ok = true; break; }
                    if (ok) usable.set_1(i); else usable.set_0(i);
                    ++i;
                }
            }
            if (nk.size() == 1) {                                                              // :259-267
                result out(nk[0], false, false);
                for (uint64_t i = 0; i < nk[0]->size(); ++i)
                    if (usable.at(i)) out.should_use(i);
                return out;
            }
            if (rest_n == 0) {                                                                 // :270-298
                result out(nk[0], true, false);
                for (uint64_t s = 0; s < nk.front()->size(); ++s) {
                    uint64_t prev = (*nk.front())[s];
                    bool use = true;
                    for (uint64_t j = 1; j < nk.size(); ++j) {
                        const auto* cur = nk[j];
                        if (!std::binary_search(cur->begin(), cur->end(), uint32_t(prev + k))) {
                            out.should_not_use(s); use = false; break;
                        }
                        prev += k;
                    }
                    if (use) out.should_use(s);
                }
                return out;
            }
            // :301-338 — n >= 2 parts plus a rest.
            result out(nk[0], true, false);
            for (uint64_t s = 0; s < nk.front()->size(); ++s) {
                uint64_t prev = (*nk.front())[s];
                bool interrupted = false;
                for (uint64_t j = 1; j < nk.size(); ++j) {
                    // :314 reads nk_positions.back() for EVERY j (defect 1);
                    // the repaired form walks part j.
                    const auto* cur = (mode == MODE_FAITHFUL) ? nk.back() : nk[j];
                    prev += k;
                    auto it = std::lower_bound(cur->begin(), cur->end(), uint32_t(prev));
                    if (it == cur->end() || *it != prev) { interrupted = true; break; }        // :317 (UB at end() -> mismatch)
                    if (mode == MODE_FAITHFUL) {
                        if (j == nk.size() - 1) {                                              // :323-329
                            if (!usable.at(uint64_t(it - cur->begin()))) interrupted = true;
                            break;
                        }
                    } else if (j == nk.size() - 1) {
                        if (!usable.at(uint64_t(it - cur->begin()))) interrupted = true;
                    }
                }
                if (interrupted) out.should_not_use(s);
            }
            return out;
        }
        // m < k — :342-345
        return result(prefix_buckets(q, m));
    }
};

// ---------------------------------------------------------------------------
// kmer_index<alphabet_t, uint32_t, ks...> — kmer_index.hpp:350-566.
// ---------------------------------------------------------------------------
static const uint64_t QUERY_SIZE_RANGE = 10000;                                               // :401

struct planner {
    std::vector<uint64_t> all_ks;                                                              // :360
    std::vector<std::vector<uint64_t>> nk_sum;                                                 // _optimal_nk_sum :404
    std::vector<uint8_t> multi;                                                                // _use_multi_search_scheme :405

    // choose_search_scheme — :407-476
    void choose(const uint32_t* ks, uint32_t n_ks, uint64_t range)
    {
        all_ks.assign(ks, ks + n_ks);
        std::sort(all_ks.begin(), all_ks.end(), [](uint64_t a, uint64_t b) { return a > b; }); // :410
        std::vector<uint64_t> high;
        for (uint64_t k : all_ks) if (k >= 9) high.push_back(k);                               // :412-415
        nk_sum.assign(range, {});
        multi.assign(range, 0);
        for (uint64_t k : high) if (k < range) { nk_sum[k] = {k}; multi[k] = 1; }              // :421-425
        for (uint64_t q = all_ks.front() + 1; q < range; ++q) {                                // :427-443
            for (uint64_t k : high) {
                if (!nk_sum[q - k].empty()) {
                    nk_sum[q] = nk_sum[q - k];
                    nk_sum[q].push_back(k);
                    multi[q] = 1;
                    break;
                }
            }
        }
        for (uint64_t q = 0; q < range; ++q) {                                                 // :445-475
            if (!nk_sum[q].empty()) continue;
            uint64_t best = all_ks.front();
            if (q < all_ks.front()) {                                                          // :450-461
                for (uint64_t k : all_ks)
                    if (q <= k && (k - q < best - q)) best = k;
            } else {                                                                           // :463-473 (float ceil restated)
                for (uint64_t k : all_ks) {
                    float a = std::ceil(float(q) / float(k)) * float(k) - float(q);
                    float b = std::ceil(float(q) / float(best)) * float(best) - float(q);
                    if (a < b) best = k;
                }
            }
            nk_sum[q] = {best};
        }
    }
};

struct index {
    uint32_t sigma = 0;
    uint64_t n = 0;
    std::vector<uint32_t> ks;                 // template order
    std::vector<element> elems;               // one per k, template order
    planner plan;

    const element& elem_for(uint64_t k) const
    {
        for (size_t i = 0; i < ks.size(); ++i) if (ks[i] == k) return elems[i];               // _k_to_search_fns_i :387-398
        throw std::logic_error("no element for k");
    }

    // search — :505-558
    result search(const uint8_t* q, uint64_t m, int mode) const
    {
        if (m == 0) throw search_error{ST_EMPTY_QUERY};                                        // assert :195
        if (m >= QUERY_SIZE_RANGE) throw search_error{ST_TOO_LONG};                            // :507-509 (== range is an OOB read at :512)
        if (!plan.multi[m] || plan.all_ks.size() == 1)                                         // :512-513
            return elem_for(plan.nk_sum[m].at(0)).search(q, m, mode);

        const auto& sum = plan.nk_sum[m];
        std::vector<const bucket_t*> nk;
        uint64_t last_k = 0;
        for (uint64_t cur_k : sum) {                                                           // :518-527
            const auto* pos = elem_for(cur_k).search_k(q + last_k);
            if (!pos) return result();
            nk.push_back(pos);
            // :526 assigns (`last_k = current_k`) where the running offset needs `+=`
            // (defect 2); identical for <= 2 summands.
            if (mode == MODE_FAITHFUL) last_k = cur_k; else last_k += cur_k;
        }
        if (nk.size() == 1) return result(nk[0], true, true);                                  // :529-530
        result out(nk[0], true, false);                                                        // :532
        for (uint64_t s = 0; s < nk.front()->size(); ++s) {                                    // :536-555
            uint64_t prev = (*nk.front())[s];
            bool interrupted = false;
            for (uint64_t j = 1; j < nk.size(); ++j) {
                const auto* cur = nk[j];
                // :535/:544 never advance nk_sum_i, so every hop adds k_0 (defect 2).
                prev += (mode == MODE_FAITHFUL) ? sum.at(0) : sum.at(j - 1);
                auto it = std::lower_bound(cur->begin(), cur->end(), uint32_t(prev));
                if (it == cur->end() || *it != prev) { interrupted = true; break; }            // :546 (UB at end() -> mismatch)
            }
            if (interrupted) out.should_not_use(s);
        }
        return out;
    }
};

// ---------------------------------------------------------------------------
// thread_pool — thread_pool.hpp:21-109, thread_pool.cpp:12-115: FIFO queue of
// type-erased packaged tasks, execute() -> future, destructor drains the queue
// then joins.  Used by the constructor (kmer_index.hpp:485-492) and by the CPU
// batch baseline composed in SURVEY §3.3.
// ---------------------------------------------------------------------------
class thread_pool {
    std::queue<std::function<void()>> tasks_;
    std::condition_variable cv_;
    std::mutex mu_;
    std::vector<std::thread> threads_;
    bool draining_ = false;
public:
    explicit thread_pool(size_t n)
    {
        for (size_t i = 0; i < std::max<size_t>(n, 1); ++i)
            threads_.emplace_back([this] {
                for (;;) {
                    std::function<void()> task;
                    {
                        std::unique_lock<std::mutex> lock(mu_);
                        cv_.wait(lock, [this] { return draining_ || !tasks_.empty(); });      // thread_pool.cpp:24-26
                        if (tasks_.empty()) return;                                           // :37-41 (drain, then exit)
                        task = std::move(tasks_.front());                                     // :44-45
                        tasks_.pop();
                    }
                    task();
                }
            });
    }
    ~thread_pool()                                                                            // thread_pool.cpp:65-77
    {
        { std::lock_guard<std::mutex> lock(mu_); draining_ = true; }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    template <typename F>
    std::future<void> execute(F f)                                                            // thread_pool.hpp:89-108
    {
        auto task = std::make_shared<std::packaged_task<void()>>(std::move(f));
        auto fut = task->get_future();
        { std::lock_guard<std::mutex> lock(mu_); tasks_.emplace([task] { (*task)(); }); }
        cv_.notify_one();
        return fut;
    }
};

} // namespace orc

// ===========================================================================
// C entry points (ctypes / bench.py).
// ===========================================================================
extern "C" {

typedef struct orc_index orc_index;

uint64_t orc_fast_pow(uint64_t base, uint8_t exp) { return orc::fast_pow(base, exp); }

// choose_best_k — choose_best_k.hpp:12-60, restated loop for loop (equal scores keep the candidate order).
void orc_choose_best_k(const uint64_t* interval, uint64_t n, uint32_t n_k, uint32_t* out)
{
    std::vector<std::pair<uint64_t, uint64_t>> k_and_score;
    for (uint64_t k : {29, 27, 25, 23, 21, 19, 17, 13, 11, 10}) k_and_score.emplace_back(k, 0);          // :22-23
    for (uint64_t j = 0; j < n; ++j) {                                                                     // :25
        const uint64_t i = interval[j];
        for (auto& p : k_and_score) {                                                                      // :27
            const uint64_t k = p.first;
            if (i % k == 0) { p.second += 3; break; }                                                      // :32-36
            else if (k - (i % k) <= 3) { p.second += 4 - (k - (i % k)); break; }                           // :38-42
            else continue;                                                                                 // :44-45
        }
    }
    std::stable_sort(k_and_score.begin(), k_and_score.end(), [](auto a, auto b) { return a.second > b.second; });   // :50-51
    for (uint32_t i = 0; i < n_k; ++i) out[i] = uint32_t(k_and_score.at(i).first);                         // :55-57
}

// Replays a list of bit operations on a compressed_bitset and returns its words.
// ops[i] = (index << 1) | value.  Returns the word count, or -1 when an index is
// out of range (the reference throws std::out_of_range there).
int64_t orc_bitset_words(uint64_t n_bits, int fill, const uint64_t* ops, uint64_t n_ops,
                         uint64_t* words_out, uint64_t cap, uint64_t* count_ones)
{
    try {
        orc::bitset b(n_bits, fill != 0);
        for (uint64_t i = 0; i < n_ops; ++i) {
            if (ops[i] & 1) b.set_1(ops[i] >> 1); else b.set_0(ops[i] >> 1);
        }
        if (count_ones) *count_ones = b.count(true);
        for (uint64_t i = 0; i < b.words.size() && i < cap; ++i) words_out[i] = b.words[i];
        return int64_t(b.words.size());
    } catch (const std::out_of_range&) {
        return -1;
    }
}

// Planner tables.  nk_off has range+1 entries into nk_flat; returns the number
// of flat entries needed (call once with cap = 0 to size).
uint64_t orc_plan(const uint32_t* ks, uint32_t n_ks, uint64_t range, uint8_t* multi,
                  uint64_t* nk_off, uint32_t* nk_flat, uint64_t cap)
{
    orc::planner p;
    p.choose(ks, n_ks, range);
    uint64_t total = 0;
    for (uint64_t q = 0; q < range; ++q) {
        if (multi) multi[q] = p.multi[q];
        if (nk_off) nk_off[q] = total;
        for (uint64_t k : p.nk_sum[q]) {
            if (nk_flat && total < cap) nk_flat[total] = uint32_t(k);
            ++total;
        }
    }
    if (nk_off) nk_off[range] = total;
    return total;
}

orc_index* orc_build(const uint8_t* ranks, uint64_t n, uint32_t sigma, const uint32_t* ks,
                     uint32_t n_ks, uint32_t n_threads)
{
    auto* idx = new orc::index();
    idx->sigma = sigma;
    idx->n = n;
    idx->ks.assign(ks, ks + n_ks);
    idx->elems.resize(n_ks);
    for (uint32_t i = 0; i < n_ks; ++i) { idx->elems[i].sigma = sigma; idx->elems[i].k = ks[i]; }
    {
        // kmer_index.hpp:485-492 — one create() task per k on the pool, then join.
        orc::thread_pool pool(std::max<uint32_t>(n_threads, 1));
        std::vector<std::future<void>> futs;
        for (uint32_t i = 0; i < n_ks; ++i)
            futs.push_back(pool.execute([idx, i, ranks, n] { idx->elems[i].create(ranks, n); }));
        for (auto& f : futs) f.get();
    }
    idx->plan.choose(ks, n_ks, orc::QUERY_SIZE_RANGE);                                         // :494-495
    return reinterpret_cast<orc_index*>(idx);
}

void orc_free(orc_index* p) { delete reinterpret_cast<orc::index*>(p); }
void orc_free_buf(void* p) { free(p); }

// One query.  *positions (malloc'd) receives search(q).to_vector().  When
// mask_words != NULL it receives the result's bitmask words (malloc'd) and
// *mask_bits / *bypass / *n_candidates describe it.
int orc_search(const orc_index* p, const uint8_t* q, uint64_t m, int mode, uint32_t** positions,
               uint64_t* n_positions, uint64_t** mask_words, uint64_t* mask_bits, int* bypass,
               uint64_t* n_candidates)
{
    const auto* idx = reinterpret_cast<const orc::index*>(p);
    *positions = nullptr; *n_positions = 0;
    if (mask_words) { *mask_words = nullptr; *mask_bits = 0; *bypass = 0; *n_candidates = 0; }
    try {
        orc::result r = idx->search(q, m, mode);
        auto v = r.to_vector();
        *n_positions = v.size();
        *positions = static_cast<uint32_t*>(malloc(std::max<size_t>(v.size(), 1) * sizeof(uint32_t)));
        std::copy(v.begin(), v.end(), *positions);
        if (mask_words) {
            *mask_bits = r.mask.n_bits;
            *bypass = r.bypass;
            *n_candidates = r.n_results;
            *mask_words = static_cast<uint64_t*>(malloc(r.mask.words.size() * sizeof(uint64_t)));
            std::copy(r.mask.words.begin(), r.mask.words.end(), *mask_words);
        }
        return orc::ST_OK;
    } catch (const orc::search_error& e) {
        return e.status;
    }
}

// Batch search on the oracle's thread pool — the harness composed in SURVEY §3.3:
// queries are cut into >= 4*T contiguous chunks, one pool task per chunk, each task
// runs search(q).to_vector() per query.  Pass 1 keeps the vectors, then hit_off
// (nq+1) and one concatenated positions array (malloc'd) are produced.
// keep_hits = 0 discards the vectors after folding them into *checksum (timing leg).
// `runner` (may be NULL): an external pool that carries the chunk tasks — oracle/_ref's ref_pool_run, i.e. the REFERENCE's own
// thread_pool.{hpp,cpp} compiled from its sources; NULL = the restated pool above.
typedef void (*orc_pool_runner)(uint32_t n_threads, uint32_t n_tasks, void (*fn)(void*, uint32_t), void* ctx);
int orc_search_batch_on(const orc_index* p, const uint8_t* qranks, const uint64_t* qoff, uint64_t nq,
                        int mode, uint32_t n_threads, int keep_hits, uint64_t* hit_off,
                        uint32_t** positions, int32_t* status, uint64_t* checksum, orc_pool_runner runner)
{
    const auto* idx = reinterpret_cast<const orc::index*>(p);
    std::vector<std::vector<uint32_t>> hits(keep_hits ? nq : 0);
    std::vector<uint64_t> counts(nq, 0);
    uint32_t T = std::max<uint32_t>(n_threads, 1);
    uint64_t n_chunks = std::max<uint64_t>(1, std::min<uint64_t>(nq, uint64_t(4) * T));
    std::vector<uint64_t> sums(n_chunks, 0);
    auto chunk = [&](uint64_t c) {
        const uint64_t b = nq * c / n_chunks, e = nq * (c + 1) / n_chunks;
        uint64_t sum = 0;
        for (uint64_t i = b; i < e; ++i) {
            int st = orc::ST_OK;
            std::vector<uint32_t> v;
            try {
                v = idx->search(qranks + qoff[i], qoff[i + 1] - qoff[i], mode).to_vector();
            } catch (const orc::search_error& err) {
                st = err.status;
            }
            if (status) status[i] = st;
            counts[i] = v.size();
            for (uint32_t x : v) sum = sum * 1099511628211ull + x + 1;
            if (keep_hits) hits[i] = std::move(v);
        }
        sums[c] = sum;
    };
    if (runner) {
        using chunk_t = decltype(chunk);
        runner(T, uint32_t(n_chunks), [](void* ctx, uint32_t c) { (*static_cast<chunk_t*>(ctx))(c); }, &chunk);
    } else {
        orc::thread_pool pool(T);
        std::vector<std::future<void>> futs;
        for (uint64_t c = 0; c < n_chunks; ++c) futs.push_back(pool.execute([&chunk, c] { chunk(c); }));
        for (auto& f : futs) f.get();
    }
    uint64_t total = 0;
    for (uint64_t i = 0; i < nq; ++i) { if (hit_off) hit_off[i] = total; total += counts[i]; }
    if (hit_off) hit_off[nq] = total;
    if (checksum) { uint64_t s = 0; for (uint64_t c = 0; c < n_chunks; ++c) s ^= sums[c] + c; *checksum = s; }
    if (keep_hits && positions) {
        *positions = static_cast<uint32_t*>(malloc(std::max<uint64_t>(total, 1) * sizeof(uint32_t)));
        uint64_t o = 0;
        for (uint64_t i = 0; i < nq; ++i) { std::copy(hits[i].begin(), hits[i].end(), *positions + o); o += hits[i].size(); }
    }
    return 0;
}

int orc_search_batch(const orc_index* p, const uint8_t* qranks, const uint64_t* qoff, uint64_t nq,
                     int mode, uint32_t n_threads, int keep_hits, uint64_t* hit_off,
                     uint32_t** positions, int32_t* status, uint64_t* checksum)
{
    return orc_search_batch_on(p, qranks, qoff, nq, mode, n_threads, keep_hits, hit_off, positions, status, checksum, nullptr);
}

// Ground truth: every offset p with text[p..p+m) == q, ascending (what the
// reference's own test compares search(q).to_vector() with, test_main.cpp:37-45).
uint64_t orc_naive_scan(const uint8_t* text, uint64_t n, const uint8_t* q, uint64_t m,
                        uint32_t* out, uint64_t cap)
{
    uint64_t cnt = 0;
    if (m == 0 || m > n) return 0;
    for (uint64_t p = 0; p + m <= n; ++p) {
        if (text[p] != q[0]) continue;
        if (memcmp(text + p, q, m) == 0) { if (cnt < cap) out[cnt] = uint32_t(p); ++cnt; }
    }
    return cnt;
}

// Many queries of one length against one text via a rolling polynomial hash
// (exact: verified by memcmp), used to ground-truth large randomized batches.
uint64_t orc_naive_batch(const uint8_t* text, uint64_t n, const uint8_t* qranks, const uint64_t* qoff,
                         uint64_t nq, uint64_t* hit_off, uint32_t** positions)
{
    std::vector<std::vector<uint32_t>> hits(nq);
    // group queries by length
    std::unordered_map<uint64_t, std::vector<uint64_t>> by_len;
    for (uint64_t i = 0; i < nq; ++i) by_len[qoff[i + 1] - qoff[i]].push_back(i);
    const uint64_t B = 1000003ull;
    for (auto& kv : by_len) {
        uint64_t m = kv.first;
        if (m == 0 || m > n) continue;
        std::unordered_map<uint64_t, std::vector<uint64_t>> table;
        for (uint64_t qi : kv.second) {
            uint64_t h = 0;
            for (uint64_t j = 0; j < m; ++j) h = h * B + qranks[qoff[qi] + j] + 1;
            table[h].push_back(qi);
        }
        uint64_t top = 1;
        for (uint64_t j = 1; j < m; ++j) top *= B;
        uint64_t h = 0;
        for (uint64_t j = 0; j < m; ++j) h = h * B + text[j] + 1;
        for (uint64_t p = 0;; ++p) {
            auto it = table.find(h);
            if (it != table.end())
                for (uint64_t qi : it->second)
                    if (memcmp(text + p, qranks + qoff[qi], m) == 0) hits[qi].push_back(uint32_t(p));
            if (p + m >= n) break;
            h = (h - (uint64_t(text[p]) + 1) * top) * B + text[p + m] + 1;
        }
    }
    uint64_t total = 0;
    for (uint64_t i = 0; i < nq; ++i) { hit_off[i] = total; total += hits[i].size(); }
    hit_off[nq] = total;
    *positions = static_cast<uint32_t*>(malloc(std::max<uint64_t>(total, 1) * sizeof(uint32_t)));
    uint64_t o = 0;
    for (uint64_t i = 0; i < nq; ++i) { std::copy(hits[i].begin(), hits[i].end(), *positions + o); o += hits[i].size(); }
    return total;
}

} // extern "C"
