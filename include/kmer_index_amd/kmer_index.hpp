// kmer::kmer_index<alphabet_t, position_t, ks...> / kmer::make_kmer_index<ks...>() — host mirror of
// the reference's user API (kmer_index.hpp:350-579) on top of the C-ABI in include/kmx.h.
//
//   auto index = kmer::make_kmer_index<8, 10, 12>(text);                 // kmer_index.hpp:569-579
//   auto hits  = index.search(query).to_vector();                        // :505-558 + result :244-260
//   auto many  = index.search(std::vector<std::vector<dna4>>{...});      // batch overload (new)
//
// Same names, same argument meaning, same error behaviour: std::invalid_argument for a query that
// is too long (:507-509) or whose sub-k fan-out exceeds 1e7 (:119-122).  All searching happens on
// the GPU; there is no host search path.  Link with kmer_index_amd/libkmx.so.
#pragma once
#include <cstdint>
#include <memory>
#include <mutex>
#include <ranges>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../kmx.h"
#include "alphabet.hpp"
#include "kmer_index_result.hpp"

namespace kmer
{
    namespace detail
    {
        inline void throw_on(kmx_status st, const char* what)
        {
            if (st == KMX_OK) return;
            std::string msg = std::string(what) + ": " + kmx_last_error();
            if (st == KMX_ERR_INVALID_ARGUMENT || st == KMX_ERR_TOO_LARGE) throw std::invalid_argument(msg);
            throw std::runtime_error(msg);
        }

        struct result_deleter { void operator()(kmx_result* r) const { kmx_result_free(r); } };
        struct index_deleter { void operator()(kmx_index* i) const { kmx_index_free(i); } };
    } // namespace detail

    namespace detail
    {
        // kmer_index_element<alphabet_t, position_t, k> (kmer_index.hpp:39-347) as far as callers reach it: search_k.
        // The reference hands out `const std::vector<position_t>*` — a borrowed pointer to the bucket's vector inside the index,
        // nullptr when the text does not hold the k-mer (kmer_index.hpp:183-190).  The flattened index keeps buckets as runs of
        // one host arena, so the vector a caller is pointed to is made from the run the first time that k-mer is asked for and
        // then kept by the element (stable address, valid as long as the index, one copy per distinct k-mer ever asked for).
        // No GPU round trip: kmx_index_bucket_host answers from the host arena.
        template<typename alphabet_t, typename position_t, std::size_t k>
        class kmer_index_element
        {
            const kmx_index* _kmx = nullptr;
            mutable std::mutex _mu;
            mutable std::unordered_map<const std::uint32_t*, std::vector<position_t>> _buckets;

        protected:
            void bind(const kmx_index* index) { _kmx = index; }

        public:
            kmer_index_element() = default;
            kmer_index_element(kmer_index_element&& o) noexcept : _kmx(o._kmx), _buckets(std::move(o._buckets)) {}
            kmer_index_element& operator=(kmer_index_element&& o) noexcept { _kmx = o._kmx; _buckets = std::move(o._buckets); return *this; }

            // kmer_index.hpp:183-190 — at(hash(it)): the positions of the k-mer that starts at `it`, or nullptr
            template<typename iterator_t>
            const std::vector<position_t>* search_k(iterator_t it) const
            {
                std::uint8_t ranks[k];
                for (std::size_t i = 0; i < k; ++i, ++it) ranks[i] = alphabet_traits<alphabet_t>::to_rank(*it);
                const std::uint32_t* run = nullptr;
                std::uint32_t count = 0;
                throw_on(kmx_index_bucket_host(_kmx, std::uint32_t(k), ranks, &run, &count), "search_k");
                if (run == nullptr) return nullptr;
                std::lock_guard<std::mutex> lock(_mu);
                auto found = _buckets.find(run);
                if (found == _buckets.end()) found = _buckets.emplace(run, std::vector<position_t>(run, run + count)).first;
                return &found->second;                               // (node-based map: the address stays)
            }
        };
    } // namespace detail

    template<typename alphabet_t, typename position_t, std::size_t... ks>
    class kmer_index : public detail::kmer_index_element<alphabet_t, position_t, ks>...
    {
        static_assert(sizeof...(ks) > 0 && sizeof...(ks) <= KMX_MAX_KS, "between 1 and KMX_MAX_KS values of k");
        static_assert(std::is_same_v<position_t, std::uint32_t>, "the engine stores positions as uint32_t (make_kmer_index, kmer_index.hpp:575)");
        static_assert((detail::k_is_valid(detail::alphabet_traits<alphabet_t>::size, ks) && ...),
                      "the hashspace for the current k cannot be represented with only a 64-bit integer. Please specify a valid k");

        using traits = detail::alphabet_traits<alphabet_t>;
        std::unique_ptr<kmx_index, detail::index_deleter> _index;
        const std::uint32_t* _arena = nullptr;
        std::size_t _query_size_range = KMX_QUERY_SIZE_RANGE;

        void bind_elements() { (detail::kmer_index_element<alphabet_t, position_t, ks>::bind(_index.get()), ...); }

        explicit kmer_index(kmx_index* adopted) : _index(adopted)
        {
            std::uint64_t n_elems = 0;
            detail::throw_on(kmx_index_arena_host(adopted, &_arena, &n_elems), "kmer_index");
            bind_elements();
        }

    public:
        // how much of the reference's result object a search brings along (see kmer_index_result.hpp, LAZY MASKS)
        enum class mask_mode { lazy, eager, none };

    private:
        mask_mode _mask_mode = mask_mode::lazy;

        // the queries of one batch, kept for the results that may still ask for their candidate run + mask words
        struct batch_masks final : detail::mask_source<position_t>
        {
            const kmx_index* index = nullptr;
            const std::uint32_t* arena = nullptr;
            std::vector<std::uint8_t> ranks;
            std::vector<std::uint64_t> off;

            void fetch(std::size_t id, const position_t*& candidates, std::size_t& n_candidates, std::vector<std::uint64_t>& words) const override
            {
                candidates = nullptr; n_candidates = 0; words.clear();
                const std::uint64_t one_off[2] = {0, off[id + 1] - off[id]};
                kmx_result* raw = nullptr;
                detail::throw_on(kmx_search_batch(index, ranks.data() + off[id], one_off, 1, KMX_SEARCH_KEEP_MASKS, &raw), "search (mask fetch)");
                std::unique_ptr<kmx_result, detail::result_deleter> hold(raw);
                const std::uint64_t* hit_off; const std::uint32_t* positions; const std::uint8_t* status; const std::uint8_t* kinds;
                detail::throw_on(kmx_result_view(raw, &hit_off, &positions, &status, &kinds), "search (mask fetch)");
                if (status[0] != KMX_Q_OK || kinds[0] != KMX_KIND_STITCH) return;         // the reference's default result (:204,224,524)
                const std::uint64_t* mask_base; const std::uint64_t* mask_words; const std::uint32_t* cand_count; const std::uint64_t* cand_src;
                detail::throw_on(kmx_result_masks(raw, &mask_base, &mask_words, &cand_count, &cand_src), "search (mask fetch)");
                candidates = arena + cand_src[0];
                n_candidates = cand_count[0];
                words.assign(mask_words + mask_base[0], mask_words + mask_base[0] + (n_candidates / 64 + 1));
            }
        };

    public:
        using result_t = detail::kmer_index_result<position_t>;
        template<std::size_t k>
        using index_element_t = detail::kmer_index_element<alphabet_t, position_t, k>;

        kmer_index(kmer_index&& o) noexcept
            : index_element_t<ks>(std::move(static_cast<index_element_t<ks>&>(o)))..., _index(std::move(o._index)), _arena(o._arena),
              _query_size_range(o._query_size_range), _mask_mode(o._mask_mode)
        {}

        // kmer_index.hpp:480-496 — one flattened element per k (built on n_threads host threads) + planner
        // `devices` (new): the GPUs to replicate the index on; empty = the current device, or every device the
        // environment variable KMX_DEVICES names ("all" / "0,1,..."), so that an unchanged caller of
        // make_kmer_index<ks...>(text) can be given the whole node.  Batch searches shard over the replicas.
        template<std::ranges::range text_t>
        kmer_index(text_t& text, std::size_t n_threads = std::max(std::thread::hardware_concurrency(), 1u), const std::vector<int>& devices = {})
        {
            std::vector<std::uint8_t> ranks;
            ranks.reserve(std::ranges::size(text));
            for (auto const& l : text) ranks.push_back(traits::to_rank(l));
            const std::uint32_t k_arr[] = {std::uint32_t(ks)...};
            kmx_options opts{};
            opts.struct_size = sizeof(kmx_options);
            opts.device = -1;
            opts.n_threads = std::uint32_t(n_threads);
            opts.keep_host_arena = 1;
            if (devices.size() > KMX_MAX_DEVICES) throw std::invalid_argument("kmer_index: more than KMX_MAX_DEVICES devices");
            opts.n_devices = std::uint32_t(devices.size());
            for (std::size_t i = 0; i < devices.size(); ++i) opts.devices[i] = devices[i];
            if (!devices.empty()) opts.device = devices[0];
            kmx_index* raw = nullptr;
            detail::throw_on(kmx_index_build(ranks.data(), ranks.size(), std::uint32_t(traits::size), k_arr, sizeof...(ks), &opts, &raw),
                             "kmer_index");
            _index.reset(raw);
            std::uint64_t n_elems = 0;
            detail::throw_on(kmx_index_arena_host(raw, &_arena, &n_elems), "kmer_index");
            bind_elements();
        }

        // Build once, load many: the flattened image on disk (the thesis' stated intent,
        // thesis/content/02_implementation.tex:44-46).  The ks of the image must be this type's ks.
        void save(const std::string& path) const
        {
            detail::throw_on(kmx_index_save(_index.get(), path.c_str()), "kmer_index::save");
        }

        static kmer_index load(const std::string& path)
        {
            kmx_options opts{};
            opts.struct_size = sizeof(kmx_options);
            opts.device = -1;
            opts.keep_host_arena = 1;
            kmx_index* raw = nullptr;
            detail::throw_on(kmx_index_load(path.c_str(), &opts, &raw), "kmer_index::load");
            kmer_index out(raw);
            std::uint32_t sigma = 0, n_ks = 0, file_ks[KMX_MAX_KS] = {};
            detail::throw_on(kmx_index_info(raw, nullptr, &sigma, &n_ks, file_ks, nullptr, nullptr), "kmer_index::load");
            const std::uint32_t want[] = {std::uint32_t(ks)...};
            bool same = sigma == traits::size && n_ks == sizeof...(ks);
            for (std::uint32_t i = 0; same && i < n_ks; ++i) same = file_ks[i] == want[i];
            if (!same) throw std::invalid_argument("kmer_index::load: the image was built for another alphabet or other ks");
            return out;
        }

        // kmer_index.hpp:498-502
        void extend_query_size_range(std::size_t new_maximum)
        {
            detail::throw_on(kmx_index_extend_query_size_range(_index.get(), std::uint32_t(new_maximum)), "extend_query_size_range");
            _query_size_range = new_maximum;
        }

        // batch search: one GPU pass over all queries; results share the batch's buffers.
        // Per-query errors do not cost the batch: `status_out` receives one kmx_query_status per query and a query that
        // the reference would throw for (too long :507-509, sub-k fan-out :119-122, empty :195) gets an empty result.
        std::vector<result_t> search(const std::vector<std::vector<alphabet_t>>& queries, std::vector<std::uint8_t>& status_out) const
        {
            std::vector<std::uint8_t> ranks;
            std::vector<std::uint64_t> off(queries.size() + 1, 0);
            for (std::size_t i = 0; i < queries.size(); ++i) off[i + 1] = off[i] + queries[i].size();
            ranks.reserve(off.back());
            for (auto const& q : queries)
                for (auto const& l : q) ranks.push_back(traits::to_rank(l));

            const bool eager = _mask_mode == mask_mode::eager;
            kmx_result* raw = nullptr;
            detail::throw_on(kmx_search_batch(_index.get(), ranks.data(), off.data(), queries.size(),
                                              eager ? KMX_SEARCH_KEEP_MASKS : KMX_SEARCH_DEFAULT, &raw), "search");
            std::shared_ptr<kmx_result> handle(raw, detail::result_deleter{});
            const std::uint64_t* hit_off; const std::uint32_t* positions; const std::uint8_t* status; const std::uint8_t* kinds;
            detail::throw_on(kmx_result_view(raw, &hit_off, &positions, &status, &kinds), "search");
            const std::uint64_t* mask_base = nullptr; const std::uint64_t* mask_words = nullptr;
            const std::uint32_t* cand_count = nullptr; const std::uint64_t* cand_src = nullptr;
            if (eager) detail::throw_on(kmx_result_masks(raw, &mask_base, &mask_words, &cand_count, &cand_src), "search");

            status_out.assign(status, status + queries.size());
            std::vector<result_t> out;
            out.reserve(queries.size());
            std::shared_ptr<batch_masks> lazy;                     // made for the first result that may want it
            for (std::size_t i = 0; i < queries.size(); ++i)
            {
                const position_t* hits = positions ? positions + hit_off[i] : nullptr;
                const std::size_t n_hits = std::size_t(hit_off[i + 1] - hit_off[i]);
                const bool exact_or_prefix = status[i] == KMX_Q_OK && (kinds[i] == KMX_KIND_EXACT || kinds[i] == KMX_KIND_PREFIX);
                if (status[i] != KMX_Q_OK)
                    out.emplace_back();
                else if (exact_or_prefix || _mask_mode == mask_mode::none)
                {
                    if (kinds[i] == KMX_KIND_NONE) out.emplace_back();
                    else out.emplace_back(handle, hits, n_hits);
                }
                else if (eager)
                {
                    if (kinds[i] == KMX_KIND_STITCH)
                        out.emplace_back(handle, hits, n_hits, _arena + cand_src[i], std::size_t(cand_count[i]), mask_words + mask_base[i]);
                    else
                        out.emplace_back();
                }
                else
                {
                    // a cross-referenced query, or one without a hit (on the reference's planner table it may still own a candidate
                    // run whose bits are all clear): whatever the reference's result object holds is fetched on first use
                    if (!lazy)
                    {
                        lazy = std::make_shared<batch_masks>();
                        lazy->index = _index.get();
                        lazy->arena = _arena;
                    }
                    out.emplace_back(handle, hits, n_hits, lazy, i);
                }
            }
            if (lazy) { lazy->ranks = std::move(ranks); lazy->off = std::move(off); }
            return out;
        }

        // the same, with the reference's error behaviour: std::invalid_argument (a batch_query_error that also carries
        // the index of the first offending query, every query's status and the results of the others)
        struct batch_query_error : std::invalid_argument
        {
            std::size_t query_index;
            std::vector<std::uint8_t> status;
            std::vector<result_t> results;
            batch_query_error(const std::string& what, std::size_t i, std::vector<std::uint8_t> st, std::vector<result_t> res)
                : std::invalid_argument(what), query_index(i), status(std::move(st)), results(std::move(res)) {}
        };

        std::vector<result_t> search(const std::vector<std::vector<alphabet_t>>& queries) const
        {
            std::vector<std::uint8_t> status;
            std::vector<result_t> out = search(queries, status);
            for (std::size_t i = 0; i < status.size(); ++i)
            {
                std::string what;
                switch (status[i])
                {
                    case KMX_Q_OK: continue;
                    case KMX_Q_TOO_LONG:       // kmer_index.hpp:507-509
                        what = "query size exceed the maximum size " + std::to_string(_query_size_range) + " specified"; break;
                    case KMX_Q_SUBK_FANOUT:    // kmer_index.hpp:119-122
                        what = "query size too low for specified k"; break;
                    case KMX_Q_EMPTY_QUERY:    // assert(query.size() > 0), kmer_index.hpp:195
                        what = "query must not be empty"; break;
                    default:
                        what = "query holds a letter outside the alphabet"; break;
                }
                throw batch_query_error(what, i, std::move(status), std::move(out));
            }
            return out;
        }

        // kmer_index.hpp:505-558
        result_t search(std::vector<alphabet_t>& query) const
        {
            return std::move(search(std::vector<std::vector<alphabet_t>>{query}).front());
        }

        // kmer_index.hpp:561-565 (the reference forgets the return)
        result_t search(std::vector<alphabet_t>&& query) const
        {
            auto hold = std::move(query);
            return search(hold);
        }

        // kmer_index_element<.., k>::search_k (kmer_index.hpp:183-190): the bucket of the k-mer starting at `it`.
        // The reference hands out a borrowed pointer to the bucket's vector (nullptr on a miss); here the bucket
        // arrives as a result whose hits are that bucket (empty on a miss).
        template<std::size_t k, typename iterator_t>
        result_t search_k(iterator_t it) const
        {
            static_assert(((k == ks) || ...), "search_k<k>: the index holds no element for this k");
            std::vector<alphabet_t> kmer(it, it + k);
            return search(kmer);
        }

        // ... and the reference's own shape (kmer_index.hpp:183-190, reached through the public inheritance of :352): a borrowed
        // `const std::vector<position_t>*` into the index, nullptr on a miss, no GPU round trip.  On an index with ONE k,
        // index.search_k(it) is that element's; with several, name the element as the reference does internally (:388):
        // index.template element<k>().search_k(it).
        using index_element_t<ks>::search_k...;
        template<std::size_t k>
        const index_element_t<k>& element() const
        {
            static_assert(((k == ks) || ...), "element<k>: the index holds no element for this k");
            return static_cast<const index_element_t<k>&>(*this);
        }

        const kmx_index* handle() const { return _index.get(); }

        // The reference's result object carries the first part's bucket and a compressed_bitset over it for cross-referenced
        // queries (kmer_index_result.hpp:18-24).  Three ways to have it:
        //   lazy  (default) searches ask for hit lists only — nothing but those crosses PCIe, and the engine answers a single-k
        //         query longer than its k from the largest k that fits it (kmx.h, KMX_SEARCH_REFERENCE_PLAN: the same lists,
        //         several times the rate on indexes with a small k); a result fetches bucket + bits on the first call of
        //         should_use / should_not_use / is_valid / bitmask / candidates by re-issuing its one query;
        //   eager every search asks for them (KMX_SEARCH_KEEP_MASKS: the reference's planner table, mask words over PCIe);
        //   none  results of long queries bypass their mask like exact ones do (should_use / should_not_use throw
        //         std::out_of_range, as on any bypass result).
        void set_mask_mode(mask_mode m) { _mask_mode = m; }
        mask_mode get_mask_mode() const { return _mask_mode; }
        void keep_masks(bool keep) { _mask_mode = keep ? mask_mode::eager : mask_mode::none; }
        bool keeps_masks() const { return _mask_mode == mask_mode::eager; }

        // the devices this index is replicated on
        std::vector<int> devices() const
        {
            std::uint32_t n = 0;
            std::int32_t d[KMX_MAX_DEVICES] = {};
            detail::throw_on(kmx_index_devices(_index.get(), &n, d), "devices");
            return std::vector<int>(d, d + n);
        }
    };

    // choose_best_k.hpp:12-60 — which ks to instantiate for a set of query lengths
    template<std::ranges::range range_t>
    std::vector<std::size_t> choose_best_k(range_t&& interval, std::size_t n_k = 4)
    {
        std::vector<std::uint64_t> lengths;
        for (auto v : interval) lengths.push_back(std::uint64_t(v));
        std::vector<std::uint32_t> ks(n_k);
        detail::throw_on(kmx_choose_best_k(lengths.data(), lengths.size(), std::uint32_t(n_k), ks.data()), "choose_best_k");
        return std::vector<std::size_t>(ks.begin(), ks.end());
    }

    // kmer_index.hpp:569-579
    template<std::size_t... ks, std::ranges::range text_t>
    auto make_kmer_index(text_t&& text, std::size_t n_threads = std::thread::hardware_concurrency(), const std::vector<int>& devices = {})
    {
        using alphabet_t = std::remove_cvref_t<std::ranges::range_value_t<text_t>>;
        using position_t = std::uint32_t;
        return kmer_index<alphabet_t, position_t, ks...>(text, std::max<std::size_t>(n_threads, 1), devices);
    }
} // namespace kmer
