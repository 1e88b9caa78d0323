// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Thin C entry points over the parts of the REAL reference that compile from
// their own sources: fast_pow.hpp, compressed_bitset.hpp and
// thread_pool.{hpp,cpp}.  The reference files are included from where they lie
// (-I$(REF), default /root/reference); nothing is copied into this repository
// and the output goes to oracle/_ref/ only (git-ignored).
//
// kmer_index.hpp / kmer_index_result.hpp are NOT built: they include seqan3 and
// robin_hood.h, which are neither vendored in the reference nor installed in
// this image, so that part of the reference is unbuildable here.
//
// The standard headers below come first because the reference headers use
// std::vector, log2, std::out_of_range, std::mutex, std::atomic ... without
// including them (compressed_bitset.hpp:1-20, thread_pool.hpp:8-16).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <vector>

#include <fast_pow.hpp>
#include <compressed_bitset.hpp>
#include <thread_pool.hpp>

extern "C" {

uint64_t ref_fast_pow(uint64_t base, uint8_t exp) { return kmer::detail::fast_pow(base, exp); }

// Same contract as orc_bitset_words (oracle.cpp).
int64_t ref_bitset_words(uint64_t n_bits, int fill, const uint64_t* ops, uint64_t n_ops,
                         uint64_t* words_out, uint64_t cap, uint64_t* count_ones)
{
    try {
        kmer::detail::compressed_bitset<uint_fast64_t> b(n_bits, fill != 0);
        for (uint64_t i = 0; i < n_ops; ++i) {
            if (ops[i] & 1) b.set_1(ops[i] >> 1); else b.set_0(ops[i] >> 1);
        }
        if (count_ones) *count_ones = b.count_bits_equal_to(true);
        // The word vector is private: rebuild the words from at(), then append the
        // padding the constructor leaves behind (compressed_bitset.hpp:22-26): all
        // bits past n_bits keep the fill value.
        uint64_t n_words = std::max<uint64_t>(n_bits / 64 + 1, 1);
        for (uint64_t w = 0; w < n_words && w < cap; ++w) {
            uint64_t word = fill ? ~uint64_t(0) : 0;
            for (uint64_t bit = 0; bit < 64; ++bit) {
                uint64_t i = w * 64 + bit;
                if (i >= n_bits) break;
                if (b.at(i)) word |= uint64_t(1) << bit; else word &= ~(uint64_t(1) << bit);
            }
            words_out[w] = word;
        }
        return int64_t(n_words);
    } catch (const std::out_of_range&) {
        return -1;
    }
}

// Runs n_tasks tasks on the reference's thread_pool; task i adds i+1 to an atomic.
// Returns the sum (n_tasks*(n_tasks+1)/2 when execute()/future semantics hold).
uint64_t ref_pool_sum(uint32_t n_threads, uint32_t n_tasks)
{
    std::atomic<uint64_t> sum{0};
    {
        kmer::detail::thread_pool pool(n_threads);
        std::vector<std::future<void>> futs;
        for (uint32_t i = 0; i < n_tasks; ++i)
            futs.emplace_back(pool.execute([](std::atomic<uint64_t>* s, uint64_t v) { s->fetch_add(v); }, &sum, uint64_t(i + 1)));
        for (auto& f : futs) f.get();
    }
    return sum.load();
}

// Runs fn(ctx, i) for i in [0, n_tasks) as n_tasks tasks of the reference's thread_pool (execute -> future, FIFO queue,
// thread_pool.hpp:89-108 / thread_pool.cpp:12-53) and joins them: the carrier of the CPU baseline's batch
// (the reference has no batch search of its own; this is the harness a user of its pool would write).
void ref_pool_run(uint32_t n_threads, uint32_t n_tasks, void (*fn)(void*, uint32_t), void* ctx)
{
    kmer::detail::thread_pool pool(n_threads);
    std::vector<std::future<void>> futs;
    futs.reserve(n_tasks);
    for (uint32_t i = 0; i < n_tasks; ++i) futs.emplace_back(pool.execute(fn, ctx, i));
    for (auto& f : futs) f.get();
}

} // extern "C"
