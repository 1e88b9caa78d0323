// Drop-in check of the C++ host mirror: code shaped like the reference's own test
// (test_main.cpp:21-69 — build single-k and multi-k indices, compare search(q).to_vector() with the
// exact occurrence list for query sizes around k) compiled against include/kmer_index_amd/ and run
// on the GPU.  Ground truth is a naive scan (the reference uses seqan3::fm_index, absent here).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <kmer_index_amd/kmer_index.hpp>

using kmer::alphabet::aa20;
using kmer::alphabet::dna4;
using kmer::alphabet::dna15;

static std::uint64_t mix64(std::uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}

template<typename alphabet_t>
std::vector<alphabet_t> generate_sequence(std::uint64_t seed, std::size_t length)
{
    std::vector<alphabet_t> out(length);
    for (std::size_t i = 0; i < length; ++i)
        out[i].assign_rank(std::uint8_t(((mix64(seed + (i + 1) * 0x9E3779B97F4A7C15ull) >> 32) * alphabet_t::alphabet_size) >> 32));
    return out;
}

template<typename alphabet_t>
std::vector<std::uint32_t> naive(const std::vector<alphabet_t>& text, const std::vector<alphabet_t>& q)
{
    std::vector<std::uint32_t> out;
    if (q.empty() || q.size() > text.size()) return out;
    for (std::size_t p = 0; p + q.size() <= text.size(); ++p)
    {
        bool eq = true;
        for (std::size_t j = 0; j < q.size() && eq; ++j) eq = text[p + j] == q[j];
        if (eq) out.push_back(std::uint32_t(p));
    }
    return out;
}

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("CHECK failed: %s (line %d)\n", #cond, __LINE__); ++failures; } } while (0)

template<typename alphabet_t, std::size_t k>
void run_test(std::size_t text_size, std::uint64_t seed)
{
    auto text = generate_sequence<alphabet_t>(seed, text_size);
    auto single_kmer = kmer::make_kmer_index<k>(text);
    auto multi_kmer = kmer::make_kmer_index<k, k + 1, k + 2>(text);

    std::vector<std::vector<alphabet_t>> queries;
    for (std::size_t query_size = (k > 5 ? k - 5 : 1); query_size < 2 * k; query_size++)
    {
        queries.push_back(generate_sequence<alphabet_t>(seed * 131 + query_size, query_size));               // random
        std::size_t s = (seed * 7919 + query_size * 104729) % (text_size - query_size);
        queries.emplace_back(text.begin() + s, text.begin() + s + query_size);                                // planted
        queries.emplace_back(text.end() - query_size - (query_size % 3), text.end() - (query_size % 3));      // tail
    }
    // one query at a time (the reference's call shape) ...
    for (auto& query : queries)
    {
        auto truth = naive(text, query);
        CHECK(single_kmer.search(query).to_vector() == truth);
        CHECK(multi_kmer.search(query).to_vector() == truth);
    }
    // search_k (kmer_index.hpp:183-190): the bucket of one k-mer, per element of a multi-k index
    {
        std::size_t s = (seed * 31337) % (text_size - (k + 2));
        std::vector<alphabet_t> a(text.begin() + s, text.begin() + s + k), b(text.begin() + s, text.begin() + s + k + 2);
        CHECK(single_kmer.template search_k<k>(text.begin() + s).to_vector() == naive(text, a));
        CHECK(multi_kmer.template search_k<k + 2>(text.begin() + s).to_vector() == naive(text, b));
    }
    // search_k with the reference's OWN shape (kmer_index.hpp:183-190): a borrowed `const std::vector<position_t>*` into the index,
    // nullptr on a miss — used the way kmer_index::search uses it at :516-527 / :532-555 (parts looked up one by one, then the first
    // part's positions cross-referenced against the second's)
    {
        std::size_t s = (seed * 52361) % (text_size - 2 * k);
        std::vector<alphabet_t> query(text.begin() + s, text.begin() + s + 2 * k);
        std::vector<const std::vector<std::uint32_t>*> nk_positions;
        std::size_t last_k = 0;
        for (std::size_t current_k : {k, k})
        {
            const auto* pos = single_kmer.search_k(query.begin() + last_k);
            if (pos)
                nk_positions.push_back(pos);
            else
                break;
            last_k = current_k;
        }
        CHECK(nk_positions.size() == 2);
        if (nk_positions.size() == 2)
        {
            std::vector<alphabet_t> a(query.begin(), query.begin() + k), b(query.begin() + k, query.end());
            CHECK(*nk_positions[0] == naive(text, a) && *nk_positions[1] == naive(text, b));
            CHECK(single_kmer.search_k(query.begin()) == nk_positions[0]);                      // the SAME borrowed vector every time
            std::vector<std::uint32_t> stitched;
            for (std::size_t start_pos_i = 0; start_pos_i < nk_positions.front()->size(); ++start_pos_i)
            {
                const std::uint32_t previous_pos = nk_positions.front()->at(start_pos_i);
                const auto* current = nk_positions.at(1);
                auto it = std::lower_bound(current->begin(), current->end(), previous_pos + std::uint32_t(k));
                if (it != current->end() && *it == previous_pos + k) stitched.push_back(previous_pos);
            }
            CHECK(stitched == naive(text, query));
        }
        auto random_kmer = generate_sequence<alphabet_t>(seed * 977 + 5, k);
        const auto* maybe = single_kmer.search_k(random_kmer.begin());
        auto truth = naive(text, random_kmer);
        CHECK((maybe == nullptr) == truth.empty());
        if (maybe) CHECK(*maybe == truth);
        // an index with several ks: the element is named, as the reference does internally (:388)
        std::vector<alphabet_t> c(text.begin() + s, text.begin() + s + k + 2);
        const auto* pos2 = multi_kmer.template element<k + 2>().search_k(text.begin() + s);
        CHECK(pos2 != nullptr && *pos2 == naive(text, c));
        const auto* pos0 = multi_kmer.template element<k>().search_k(text.begin() + s);
        std::vector<alphabet_t> d(text.begin() + s, text.begin() + s + k);
        CHECK(pos0 != nullptr && *pos0 == naive(text, d));
    }
    // ... and the batch overload
    CHECK(multi_kmer.get_mask_mode() == decltype(multi_kmer)::mask_mode::lazy);     // the default: hit lists only, masks on demand
    auto batch = multi_kmer.search(queries);
    CHECK(batch.size() == queries.size());
    for (std::size_t i = 0; i < queries.size(); ++i)
    {
        auto truth = naive(text, queries[i]);
        CHECK(batch[i].to_vector() == truth);
        CHECK(batch[i].size() == truth.size());
        if (!truth.empty()) CHECK(batch[i].at(truth.size() - 1) == truth.back() && batch[i][0] == truth.front());
        {
            bool threw = false;
            try { (void)batch[i].at(truth.size()); } catch (const std::out_of_range&) { threw = true; }
            CHECK(threw);
        }
        std::vector<std::uint32_t> via_iter(batch[i].begin(), batch[i].end());
        CHECK(via_iter == truth);
        CHECK(batch[i].bypasses_bitmask() || !batch[i].mask_is_resident());      // nothing of the mask has crossed PCIe yet
        if (!batch[i].bypasses_bitmask())
        {
            // zero-copy view: candidates filtered by the mask == to_vector()
            std::vector<std::uint32_t> filtered;
            for (std::size_t c = 0; c < batch[i].n_candidates(); ++c)
                if (batch[i].is_valid(c)) filtered.push_back(batch[i].candidates()[c]);
            CHECK(filtered == truth);
            CHECK(batch[i].bitmask().count_bits_equal_to(true) == truth.size());
            // should_not_use / should_use are mutators (kmer_index_result.hpp:228-236): to_vector() follows the edited mask
            if (batch[i].n_candidates() > 0)
            {
                auto edited = batch[i];
                std::size_t c = 0;
                while (c + 1 < edited.n_candidates() && !edited.is_valid(c)) ++c;
                const bool was = edited.is_valid(c);
                const std::uint32_t p = edited.candidates()[c];
                edited.should_not_use(c);
                auto without = truth;
                if (was) without.erase(std::find(without.begin(), without.end(), p));
                CHECK(edited.to_vector() == without && edited.size() == without.size() && !edited.is_valid(c));
                edited.should_use(c);
                auto with = without;
                with.insert(std::lower_bound(with.begin(), with.end(), p), p);
                CHECK(edited.to_vector() == with && edited.is_valid(c));
                CHECK(batch[i].to_vector() == truth);                      // the original result is untouched
            }
        }
        else
        {
            // a result that bypasses its mask has a 0-bit mask: the reference's set_0 / set_1 throw std::out_of_range
            auto copy = batch[i];
            bool threw = false;
            try { copy.should_not_use(0); } catch (const std::out_of_range&) { threw = true; }
            CHECK(threw);
        }
    }
    // without the reference's result object (keep_masks(false)): the same lists, every result bypasses its mask
    multi_kmer.keep_masks(false);
    auto lean = multi_kmer.search(queries);
    CHECK(lean.size() == queries.size());
    for (std::size_t i = 0; i < queries.size(); ++i)
    {
        CHECK(lean[i].to_vector() == batch[i].to_vector());
        CHECK(lean[i].size() == 0 || lean[i].bypasses_bitmask());
    }
    CHECK(multi_kmer.search(queries[0]).to_vector() == batch[0].to_vector());
    // the reference's result object fetched WITH the search (keep_masks(true)): what the lazy results fetched one by one
    multi_kmer.keep_masks(true);
    auto eager = multi_kmer.search(queries);
    CHECK(eager.size() == queries.size());
    for (std::size_t i = 0; i < queries.size(); ++i)
    {
        CHECK(eager[i].to_vector() == batch[i].to_vector() && eager[i].mask_is_resident());
        CHECK(eager[i].bypasses_bitmask() == batch[i].bypasses_bitmask());
        if (!eager[i].bypasses_bitmask())
        {
            CHECK(eager[i].n_candidates() == batch[i].n_candidates() && eager[i].bitmask().words() == batch[i].bitmask().words());
            if (!(eager[i].n_candidates() == batch[i].n_candidates() && eager[i].bitmask().words() == batch[i].bitmask().words()))
                std::printf("  query %zu (m = %zu): eager %zu candidates, %zu words, first word %llx; lazy %zu candidates, %zu words, first word %llx\n", i, queries[i].size(),
                            eager[i].n_candidates(), eager[i].bitmask().words().size(), (unsigned long long)eager[i].bitmask().words()[0],
                            batch[i].n_candidates(), batch[i].bitmask().words().size(), (unsigned long long)batch[i].bitmask().words()[0]);
            CHECK(eager[i].n_candidates() == 0 || eager[i].candidates() == batch[i].candidates());      // the same run of the host arena
        }
    }
    multi_kmer.set_mask_mode(decltype(multi_kmer)::mask_mode::lazy);
}

int main()
{
    run_test<dna4, 10>(200000, 1);
    run_test<dna4, 5>(50000, 2);
    run_test<dna15, 5>(100000, 3);
    run_test<aa20, 4>(100000, 4);

    // error behaviour of the reference: std::invalid_argument (kmer_index.hpp:507-509, :119-122)
    {
        auto text = generate_sequence<dna4>(9, 100000);
        auto index = kmer::make_kmer_index<13>(text);
        bool threw = false;
        try { std::vector<dna4> q(10001); index.search(q); } catch (const std::invalid_argument&) { threw = true; }
        CHECK(threw);
        threw = false;
        try { std::vector<dna4> q(text.begin(), text.begin() + 1); index.search(q); } catch (const std::invalid_argument&) { threw = true; }
        CHECK(threw);                                   // 4^12 > 1e7 buckets
        // a batch keeps its good results when one query is bad
        {
            std::vector<std::vector<dna4>> qs;
            qs.emplace_back(text.begin() + 5, text.begin() + 7);
            qs.emplace_back(text.begin(), text.begin() + 1);             // sub-k fan-out
            qs.emplace_back(text.begin() + 40, text.begin() + 53);
            std::vector<std::uint8_t> st;
            auto rs = index.search(qs, st);
            CHECK(st.size() == 3 && st[0] == KMX_Q_OK && st[1] == KMX_Q_SUBK_FANOUT && st[2] == KMX_Q_OK);
            CHECK(rs[0].to_vector() == naive(text, qs[0]) && rs[1].empty() && rs[2].to_vector() == naive(text, qs[2]));
            bool caught = false;
            try { (void)index.search(qs); }
            catch (const decltype(index)::batch_query_error& e) { caught = e.query_index == 1 && e.results.size() == 3 && e.results[2].to_vector() == naive(text, qs[2]); }
            CHECK(caught);
        }
        std::vector<dna4> ok(text.begin() + 5, text.begin() + 7);
        CHECK(index.search(ok).to_vector() == naive(text, ok));
        CHECK(index.search(std::vector<dna4>(text.begin() + 50, text.begin() + 63)).size() >= 1);   // rvalue overload returns
        // extend_query_size_range (kmer_index.hpp:498-502)
        index.extend_query_size_range(20000);
        std::vector<dna4> longq(text.begin() + 100, text.begin() + 100 + 15000);
        auto r = index.search(longq).to_vector();
        CHECK(r.size() >= 1 && r.front() == 100);
    }
    // save / load round trip of the flattened image
    {
        auto text = generate_sequence<dna4>(12, 80000);
        auto index = kmer::make_kmer_index<7, 9>(text);
        std::vector<dna4> q(text.begin() + 321, text.begin() + 321 + 16);
        auto before = index.search(q).to_vector();
        index.save("/tmp/kmx_host_api_test.img");
        auto loaded = kmer::kmer_index<dna4, std::uint32_t, 7, 9>::load("/tmp/kmx_host_api_test.img");
        CHECK(loaded.search(q).to_vector() == before);
        CHECK(before == naive(text, q));
        bool threw = false;
        try { (void)kmer::kmer_index<dna4, std::uint32_t, 7, 10>::load("/tmp/kmx_host_api_test.img"); } catch (const std::invalid_argument&) { threw = true; }
        CHECK(threw);
        std::remove("/tmp/kmx_host_api_test.img");
    }
    // one index over several replicas (make_kmer_index's `devices`): a batch sharded over them == the single-device batch
    {
        auto text = generate_sequence<dna4>(21, 150000);
        auto one = kmer::make_kmer_index<8, 10>(text);
        auto many = kmer::make_kmer_index<8, 10>(text, 2, std::vector<int>{0, 0, 0});
        CHECK(many.devices().size() == 3 && one.devices().size() == 1);
        std::vector<std::vector<dna4>> qs;
        for (std::size_t i = 0; i < 200; ++i)
        {
            const std::size_t len = 5 + (i * 7) % 26, s = (i * 7919) % (text.size() - len);
            qs.emplace_back(text.begin() + s, text.begin() + s + len);
        }
        auto a = one.search(qs), b = many.search(qs);
        CHECK(a.size() == b.size());
        for (std::size_t i = 0; i < qs.size(); ++i)
        {
            CHECK(a[i].to_vector() == b[i].to_vector() && a[i].to_vector() == naive(text, qs[i]));
            CHECK(a[i].bypasses_bitmask() == b[i].bypasses_bitmask());
            if (!a[i].bypasses_bitmask()) CHECK(a[i].bitmask().words() == b[i].bitmask().words() && a[i].n_candidates() == b[i].n_candidates());
        }
        CHECK(many.search(qs[3]).to_vector() == naive(text, qs[3]));       // a batch of one lands on the first replica
    }
    if (failures) { std::printf("%d failure(s)\n", failures); return 1; }
    std::printf("host api ok\n");
    return 0;
}
