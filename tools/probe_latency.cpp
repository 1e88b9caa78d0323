// Latency of kmer_index::search(query).to_vector() through the C++ host mirror — the reference's call shape
// (test_main.cpp:41-42): a batch of one.  Build + run: python tools/probe_latency_cpp.py
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#include <kmer_index_amd/kmer_index.hpp>

using kmer::alphabet::dna4;

static std::uint64_t mix64(std::uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}

int main()
{
    const std::size_t n = 10000000;
    std::vector<dna4> text(n);
    for (std::size_t i = 0; i < n; ++i) text[i].assign_rank(std::uint8_t(((mix64(1002 + (i + 1) * 0x9E3779B97F4A7C15ull) >> 32) * 4) >> 32));
    auto index = kmer::make_kmer_index<8, 10, 12>(text);
    struct Case { const char* name; std::size_t len; };
    for (Case c : {Case{"exact  m=10", 10}, Case{"exact  m=12", 12}, Case{"prefix m=9 ", 9}, Case{"stitch m=22", 22}, Case{"stitch m=31", 31}})
    {
        std::vector<double> us;
        std::size_t hits = 0;
        for (int rep = 0; rep < 300; ++rep)
        {
            const std::size_t s = (std::size_t(rep) * 7919 * 131) % (n - c.len);
            std::vector<dna4> q(text.begin() + s, text.begin() + s + c.len);
            const auto t0 = std::chrono::steady_clock::now();
            auto v = index.search(q).to_vector();
            const auto t1 = std::chrono::steady_clock::now();
            hits += v.size();
            if (rep >= 20) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::sort(us.begin(), us.end());
        std::printf("search(q).to_vector() %s: median %7.1f us  p10 %7.1f  p90 %7.1f  (%.1f hits/query)\n", c.name, us[us.size() / 2],
                    us[us.size() / 10], us[us.size() * 9 / 10], double(hits) / 300);
    }
    // 13-letter reads on {8, 10, 12} through the mirror's batch overload: the DEFAULT (lazy masks: hit lists only, engine planner
    // table) against keep_masks(true) (the reference's result object with every search: its planner table, mask words over PCIe)
    {
        const std::size_t nq = 500000, len = 13;
        std::vector<std::vector<dna4>> reads;
        reads.reserve(nq);
        for (std::size_t i = 0; i < nq; ++i)
        {
            const std::size_t s = (i * 7919 * 131) % (n - len);
            reads.emplace_back(text.begin() + s, text.begin() + s + len);
        }
        for (int mode = 0; mode < 2; ++mode)
        {
            if (mode == 1) index.keep_masks(true);
            std::size_t hits = 0;
            double best = 1e30;
            for (int rep = 0; rep < 4; ++rep)
            {
                const auto t0 = std::chrono::steady_clock::now();
                auto res = index.search(reads);
                const auto t1 = std::chrono::steady_clock::now();
                hits = 0;
                for (auto const& r : res) hits += r.size();
                if (rep) best = std::min(best, std::chrono::duration<double>(t1 - t0).count());
            }
            std::printf("search(batch of %zu 13-letter reads) %s: %7.1f M queries/s end to end (host vectors in, result views out; %.2f hits/query)\n", nq,
                        mode == 0 ? "default (lazy masks, engine plan)   " : "keep_masks(true) (reference's plan) ", double(nq) / best / 1e6, double(hits) / double(nq));
        }
        index.set_mask_mode(decltype(index)::mask_mode::lazy);
    }
    return 0;
}
