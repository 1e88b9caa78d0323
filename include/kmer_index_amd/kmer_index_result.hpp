// kmer::detail::kmer_index_result — host mirror of the reference's result view
// (kmer_index_result.hpp:15-272).
//
// Reference: pointers to the index's bucket vectors + a compressed_bitset + a bypass flag;
// to_vector() copies the valid positions and sorts them.  Here the engine has already
// materialised to_vector() on the GPU, so a result is a window into the batch's hit buffer
// (kept alive by a shared handle) and, for cross-referenced queries, the candidate run (a window
// into the index's host arena) plus the mask words.  Observable differences, all documented
// reference defects:
//   * size() returns the number of valid positions (the reference counts mask bits and so
//     returns 0 for bypass and sub-k results, kmer_index_result.hpp:239-242);
//   * begin()/end() work (the reference's iterator does not instantiate, :182).
//
// LAZY MASKS.  The reference's result of a cross-referenced query (m > k, multi-k sums) always carries the first part's
// bucket and a compressed_bitset over it (:18-24).  Almost every caller only reads to_vector() / size() / iterators, so
// by default the search asks the engine for hit lists only, and a result fetches its candidate run + mask words the FIRST time
// one of should_use / should_not_use / is_valid / bitmask / candidates / n_candidates is called — by re-issuing that one query
// with KMX_SEARCH_KEEP_MASKS (the single-launch latency path, ~20 us).  What those members then report is exactly what an
// eager search reports (kmer_index::keep_masks(true)): the reference planner's bucket and bits.  The index must outlive the
// result, as in the reference (whose results hold raw pointers into the index's buckets, :23).  The first mask access of one
// result object is not thread-safe against another access of the SAME object (copies are independent).
//
// should_use(i) / should_not_use(i) are MUTATORS as in the reference (:228-236): they set / clear bit i of the
// result's own copy of the mask, and to_vector() / size() / begin() / at() follow the edited mask from then on
// (the hits are re-derived from the candidate run, which is ascending, so the list stays sorted).  On a result that
// bypasses its mask (exact and sub-k lookups) the mask has 0 bits and both throw std::out_of_range, exactly what
// compressed_bitset::set_0 / set_1 do in the reference for such a result.  The const predicate is is_valid(i).
#pragma once
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "compressed_bitset.hpp"

namespace kmer::detail
{
    enum class BYPASS_BITMASK : bool { YES = true, NO = false };

    // where a result gets its candidate run + mask words from when they were not fetched with the search (kmer_index makes these)
    template<typename position_t>
    struct mask_source
    {
        virtual ~mask_source() = default;
        // query `id` of the batch this source belongs to: the first part's bucket (NULL / 0: the reference returns its
        // default result for this query) and n_candidates / 64 + 1 mask words
        virtual void fetch(std::size_t id, const position_t*& candidates, std::size_t& n_candidates, std::vector<std::uint64_t>& words) const = 0;
    };

    template<typename position_t>
    class kmer_index_result
    {
        std::shared_ptr<void> _keep_alive;              // the batch result handle the windows point into
        mutable const position_t* _hits = nullptr;      // ascending valid positions (= to_vector())
        mutable std::size_t _n_hits = 0;
        mutable const position_t* _candidates = nullptr; // first part's bucket (only for masked results)
        mutable std::size_t _n_candidates = 0;
        mutable compressed_bitset<std::uint_fast64_t> _bitmask; // validity over the candidates
        bool _bypass_bitmask = true;
        std::shared_ptr<std::vector<position_t>> _edited; // hits re-derived after should_use / should_not_use
        std::shared_ptr<const mask_source<position_t>> _lazy;   // set while the mask has not been fetched yet
        std::size_t _lazy_id = 0;
        mutable bool _fetched = true;

        void ensure_mask() const
        {
            if (_fetched) return;
            _fetched = true;
            std::vector<std::uint64_t> words;
            _lazy->fetch(_lazy_id, _candidates, _n_candidates, words);
            if (_candidates == nullptr)
            {
                _n_candidates = 0;                                    // the reference's default result (kmer_index.hpp:204,224,524)
                _bitmask = compressed_bitset<std::uint_fast64_t>(0, true);
                return;
            }
            if (words.size() < _n_candidates / 64 + 1) words.resize(_n_candidates / 64 + 1, 0);
            _bitmask = compressed_bitset<std::uint_fast64_t>(_n_candidates, words.data());
        }

        void rederive()
        {
            auto v = std::make_shared<std::vector<position_t>>();
            for (std::size_t i = 0; i < _n_candidates; ++i)
                if (_bitmask.at(i)) v->push_back(_candidates[i]);
            _edited = std::move(v);
            _hits = _edited->data();
            _n_hits = _edited->size();
        }

    public:
        using const_iterator = const position_t*;

        kmer_index_result() : _bitmask(0, true), _bypass_bitmask(false) {}

        // exact / sub-k result: the bucket(s) by reference, bitmask bypassed (kmer_index.hpp:198-205, :342-345, :529-530)
        kmer_index_result(std::shared_ptr<void> keep, const position_t* hits, std::size_t n_hits)
            : _keep_alive(std::move(keep)), _hits(hits), _n_hits(n_hits), _bitmask(0, true), _bypass_bitmask(true)
        {}

        // cross-referenced result with its candidate run and mask words already here (eager: KMX_SEARCH_KEEP_MASKS)
        kmer_index_result(std::shared_ptr<void> keep, const position_t* hits, std::size_t n_hits,
                          const position_t* candidates, std::size_t n_candidates, const std::uint64_t* mask_words)
            : _keep_alive(std::move(keep)), _hits(hits), _n_hits(n_hits), _candidates(candidates), _n_candidates(n_candidates),
              _bitmask(n_candidates, mask_words), _bypass_bitmask(false)
        {}

        // cross-referenced result (or a miss) whose candidate run and mask words are fetched on first use
        kmer_index_result(std::shared_ptr<void> keep, const position_t* hits, std::size_t n_hits,
                          std::shared_ptr<const mask_source<position_t>> source, std::size_t id)
            : _keep_alive(std::move(keep)), _hits(hits), _n_hits(n_hits), _bitmask(0, true), _bypass_bitmask(false),
              _lazy(std::move(source)), _lazy_id(id), _fetched(false)
        {}

        // number of valid positions
        std::size_t size() const { return _n_hits; }
        bool empty() const { return _n_hits == 0; }

        // kmer_index_result.hpp:244-260
        std::vector<position_t> to_vector() const { return std::vector<position_t>(_hits, _hits + _n_hits); }

        const_iterator begin() const { return _hits; }
        const_iterator end() const { return _hits + _n_hits; }
        // the i-th valid position (the intended `at` of ~kmer_index_result.hpp:223-241); std::out_of_range past the end
        position_t at(std::size_t i) const
        {
            if (i >= _n_hits) throw std::out_of_range("kmer_index_result::at: index " + std::to_string(i) + " out of range");
            return _hits[i];
        }
        position_t operator[](std::size_t i) const { return _hits[i]; }

        // specify which positions to use by setting the bitmask (kmer_index_result.hpp:228-236)
        void should_not_use(std::size_t i) { ensure_mask(); _bitmask.set_0(i); rederive(); }
        void should_use(std::size_t i) { ensure_mask(); _bitmask.set_1(i); rederive(); }

        // the zero-copy view of the reference: candidates + validity mask
        bool bypasses_bitmask() const { return _bypass_bitmask; }
        std::size_t n_candidates() const { if (_bypass_bitmask) return _n_hits; ensure_mask(); return _n_candidates; }
        const position_t* candidates() const { if (_bypass_bitmask) return _hits; ensure_mask(); return _candidates; }
        bool is_valid(std::size_t i) const { if (_bypass_bitmask) return true; ensure_mask(); return _bitmask.at(i); }
        const compressed_bitset<std::uint_fast64_t>& bitmask() const { ensure_mask(); return _bitmask; }
        // has the candidate run + mask been fetched (or was it never lazy)?
        bool mask_is_resident() const { return _fetched; }
    };
} // namespace kmer::detail
