// CPU check of the host side of index construction (kmx_host.cpp), compiled with -fsanitize=address,undefined by
// tests/test_host_cpp.py: the flattened image of one element — kmer_index_element::create, kmer_index.hpp:154-179 —
// against a std::map of buckets built the obvious way, for every table kind, with and without the line-aligned
// copy, histogram and sort based, including the largest valid k of an alphabet ((2, 63): key space 2^63).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "kmx_host.h"

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("CHECK failed: %s (line %d, case %s)\n", #cond, __LINE__, g_case.c_str()); ++failures; } } while (0)
static std::string g_case;

static void check_element(const std::vector<uint8_t>& text, uint32_t sigma, uint32_t k, uint32_t table, bool aligned)
{
    g_case = "sigma=" + std::to_string(sigma) + " k=" + std::to_string(k) + " n=" + std::to_string(text.size()) +
             " table=" + std::to_string(table) + " aligned=" + std::to_string(aligned);
    kmx::ElemImage im;
    std::string err;
    const bool ok = kmx::flatten_element(text.data(), text.size(), sigma, k, table, im, err, aligned);
    CHECK(ok);
    if (!ok) return;
    // ground truth: hash -> ascending positions
    std::map<uint64_t, std::vector<uint32_t>> truth;
    for (uint64_t i = 0; i + k <= text.size(); ++i) {
        uint64_t h = 0;
        for (uint32_t j = 0; j < k; ++j) h = h * sigma + text[i + j];
        truth[h].push_back(uint32_t(i));
    }
    CHECK(im.npos == text.size() - k + 1);
    CHECK(im.n_keys == kmx::key_space(sigma, k));
    CHECK(im.region >= im.npos && im.positions.size() == im.region);
    const bool has_copy = im.region > im.npos;
    if (im.table_kind == KMX_TABLE_DENSE) {
        CHECK(im.offs.size() == im.n_keys + 1);
        for (uint64_t h = 0; h < im.n_keys; ++h) {
            auto it = truth.find(h);
            const uint32_t want = it == truth.end() ? 0u : uint32_t(it->second.size());
            CHECK(im.offs[h + 1] - im.offs[h] == want);
            if (want) CHECK(std::equal(it->second.begin(), it->second.end(), im.positions.begin() + im.offs[h]));
            if (has_copy && want) {
                CHECK(im.atab.size() == im.n_keys + 1);
                const uint32_t start = im.atab[h] & ~31u;
                CHECK(start % 32 == 0 && start >= im.npos);
                CHECK(std::equal(it->second.begin(), it->second.end(), im.positions.begin() + start));
                CHECK((im.atab[h] & 31u) == (want & 31u));
            }
        }
    } else {
        CHECK(im.ukeys.size() == truth.size() && im.offs.size() == truth.size() + 1);
        uint64_t i = 0;
        for (auto& [h, ps] : truth) {
            CHECK(im.ukeys[i] == h);
            CHECK(im.offs[i + 1] - im.offs[i] == ps.size());
            CHECK(std::equal(ps.begin(), ps.end(), im.positions.begin() + im.offs[i]));
            // the probe of at(hash), kmer_index.hpp:76-84: linear probing from the multiplicative hash
            uint64_t s = kmx::slot_hash(h, im.log2cap);
            const uint64_t mask = (uint64_t(1) << im.log2cap) - 1;
            uint32_t steps = 0;
            while (im.slots[s].cnt != 0 && im.slots[s].key != h && steps++ <= mask) s = (s + 1) & mask;
            CHECK(im.slots[s].cnt == ps.size() && im.slots[s].key == h);
            CHECK(std::equal(ps.begin(), ps.end(), im.positions.begin() + im.slots[s].off));
            if (has_copy) CHECK(im.slots[s].off % 32 == 0 && im.slots[s].off >= im.npos);
            ++i;
        }
        CHECK((uint64_t(1) << im.log2cap) >= 2 * truth.size());     // load <= 0.5
        // an absent key ends on an empty slot
        uint64_t absent = truth.rbegin()->first + 1;
        if (!truth.count(absent) && absent < im.n_keys) {
            uint64_t s = kmx::slot_hash(absent, im.log2cap);
            const uint64_t mask = (uint64_t(1) << im.log2cap) - 1;
            uint32_t steps = 0;
            while (im.slots[s].cnt != 0 && im.slots[s].key != absent && steps++ <= mask) s = (s + 1) & mask;
            CHECK(im.slots[s].cnt == 0);
        }
    }
}

int main()
{
    std::mt19937_64 g(12345);
    struct Shape { uint32_t sigma, k; uint64_t n; int style; };
    const Shape shapes[] = {
        {4, 5, 3000, 0}, {4, 5, 3000, 1}, {4, 8, 60000, 0}, {4, 10, 5000, 2}, {5, 6, 20000, 0}, {20, 3, 9000, 0},
        {2, 1, 300, 0}, {2, 20, 40000, 1}, {4, 16, 30000, 1}, {4, 31, 6000, 1}, {3, 40, 6000, 1}, {2, 63, 7166, 1},
        {27, 13, 4000, 1}, {4, 12, 12, 0}, {4, 3, 3, 0}, {4, 9, 200000, 2},
    };
    for (const Shape& sh : shapes) {
        std::vector<uint8_t> text(sh.n);
        for (uint64_t i = 0; i < sh.n; ++i) {
            const uint32_t r = uint32_t(g() % sh.sigma);
            text[i] = sh.style == 0 ? r : sh.style == 1 ? (g() % 3 == 0 ? r : 0) : uint8_t((i % 37) * 7 % sh.sigma);
        }
        const uint64_t keys = kmx::key_space(sh.sigma, sh.k);
        for (uint32_t table : {uint32_t(KMX_TABLE_AUTO), uint32_t(KMX_TABLE_OPEN), uint32_t(KMX_TABLE_DENSE)}) {
            if (table == KMX_TABLE_DENSE && keys > (uint64_t(1) << 22)) continue;      // keep the truth map walk short
            for (bool aligned : {false, true}) check_element(text, sh.sigma, sh.k, table, aligned);
        }
    }
    // parameter errors (static_assert :42-43, assert :169)
    {
        kmx::ElemImage im; std::string err; std::vector<uint8_t> t(100, 0);
        g_case = "errors";
        CHECK(!kmx::flatten_element(t.data(), t.size(), 4, 32, KMX_TABLE_AUTO, im, err, true));
        CHECK(!kmx::flatten_element(t.data(), t.size(), 2, 64, KMX_TABLE_AUTO, im, err, true));
        CHECK(!kmx::flatten_element(t.data(), 5, 4, 6, KMX_TABLE_AUTO, im, err, true));
        CHECK(!kmx::flatten_element(t.data(), t.size(), 4, 0, KMX_TABLE_AUTO, im, err, true));
        CHECK(!kmx::flatten_element(t.data(), t.size(), 4, 20, KMX_TABLE_DENSE, im, err, true));
        CHECK(kmx::key_space(2, 63) == (uint64_t(1) << 63) && kmx::fast_pow(2, 63) == 0);     // fast_pow.hpp:19
    }
    // the planner (choose_search_scheme, kmer_index.hpp:407-476) and choose_best_k (choose_best_k.hpp:12-60) over random
    // k sets and ranges: every table entry is usable as the kernels use it, whatever the inputs
    {
        std::mt19937_64 prng(20261);
        for (int round = 0; round < 400; ++round) {
            std::vector<uint32_t> ks;
            const uint32_t n_ks = 1 + uint32_t(prng() % 5);
            while (ks.size() < n_ks) {
                const uint32_t k = 1 + uint32_t(prng() % 31);
                if (std::find(ks.begin(), ks.end(), k) == ks.end()) ks.push_back(k);
            }
            const uint32_t range = round < 40 ? 1 + uint32_t(round) : 1 + uint32_t(prng() % 3000);
            g_case = "plan round " + std::to_string(round) + " range " + std::to_string(range);
            const kmx::Plan p = kmx::make_plan(ks, range);
            const std::vector<KmxPlanEntry> e = kmx::make_plan_entries(ks, range);
            CHECK(p.use_multi.size() == range && p.nk_sum.size() == range && e.size() == range);
            for (uint32_t q = 0; q < range; ++q) {
                CHECK(!p.nk_sum[q].empty());
                uint64_t sum = 0;
                for (uint32_t k : p.nk_sum[q]) { sum += k; CHECK(std::find(ks.begin(), ks.end(), k) != ks.end()); }
                if (p.use_multi[q]) CHECK(sum == q);                              // a chain of summands covers the query exactly
                else CHECK(p.nk_sum[q].size() == 1);
                CHECK(e[q].elem < ks.size());
                CHECK(e[q].scheme == KMX_SCHEME_SINGLE || e[q].scheme == KMX_SCHEME_MULTI);
                if (e[q].scheme == KMX_SCHEME_MULTI) {
                    CHECK(ks.size() > 1 && e[q].nparts == p.nk_sum[q].size() && ks[e[q].elem] == p.nk_sum[q].back());
                    CHECK(q >= ks[e[q].elem]);                                    // the kernels walk q -= k down the chain
                    if (q > ks[e[q].elem]) CHECK(e[q - ks[e[q].elem]].scheme == KMX_SCHEME_MULTI || p.nk_sum[q].size() == 1);
                } else {
                    CHECK(e[q].nparts == 1);
                }
            }
            std::vector<uint64_t> lengths(prng() % 50);
            for (auto& l : lengths) l = prng() % (round % 3 == 0 ? 40 : 100000);
            for (uint32_t n_k : {0u, 1u, 3u, 10u, 25u}) {
                const std::vector<uint32_t> best = kmx::choose_best_k(lengths.empty() ? nullptr : lengths.data(), lengths.size(), n_k);
                CHECK(best.size() == std::min<size_t>(n_k, 10));
                for (size_t i = 0; i < best.size(); ++i) {
                    CHECK(best[i] >= 10 && best[i] <= 29);
                    for (size_t j = 0; j < i; ++j) CHECK(best[i] != best[j]);
                }
            }
        }
    }
    if (failures) { std::printf("%d failure(s)\n", failures); return 1; }
    std::printf("host flatten ok\n");
    return 0;
}
