#include "kmx_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace kmx {

// kmer::detail::fast_pow (fast_pow.hpp:46-93): square-and-multiply over the
// significant bits of exp; exp >= 63 is treated as overflow and yields 0 (1 when
// base == 1) — including exp == 63 itself (fast_pow.hpp:19).
uint64_t fast_pow(uint64_t base, uint8_t exp)
{
    if (exp >= 63) return base == 1 ? 1 : 0;
    uint64_t result = 1;
    while (exp) {
        if (exp & 1) result *= base;
        exp >>= 1;
        if (exp) base *= base;
    }
    return result;
}

uint64_t key_space(uint32_t sigma, uint32_t k)
{
    uint64_t v = 1;
    for (uint32_t i = 0; i < k; ++i) v *= sigma;
    return v;
}

bool k_is_valid(uint32_t sigma, uint32_t k)
{
    if (sigma < 2 || sigma > 256 || k == 0) return false;
    return double(k) < 64.0 / std::log2(double(sigma));   // kmer_index.hpp:42
}

Plan make_plan(const std::vector<uint32_t>& ks_in, uint32_t range)
{
    Plan p;
    p.use_multi.assign(range, 0);
    p.nk_sum.assign(range, {});
    std::vector<uint32_t> all_ks(ks_in);
    std::sort(all_ks.begin(), all_ks.end(), [](uint32_t a, uint32_t b) { return a > b; });   // :410
    std::vector<uint32_t> high_ks;
    for (uint32_t k : all_ks)
        if (k >= 9) high_ks.push_back(k);                                                      // :412-415
    for (uint32_t k : high_ks)                                                                 // :421-425
        if (k < range) { p.nk_sum[k] = {k}; p.use_multi[k] = 1; }
    for (uint64_t q = uint64_t(all_ks.front()) + 1; q < range; ++q) {                          // :427-443
        for (uint32_t k : high_ks) {
            if (!p.nk_sum[q - k].empty()) {
                p.nk_sum[q] = p.nk_sum[q - k];
                p.nk_sum[q].push_back(k);
                p.use_multi[q] = 1;
                break;
            }
        }
    }
    for (uint64_t q = 0; q < range; ++q) {                                                     // :445-475
        if (!p.nk_sum[q].empty()) continue;
        uint32_t best = all_ks.front();
        if (q < all_ks.front()) {
            // smallest k >= q (:450-461)
            for (uint32_t k : all_ks)
                if (q <= k && (k - q < best - q)) best = k;
        } else {
            // k with the smallest padding ceil(q/k)*k - q, first one in descending
            // order wins; the reference evaluates this in float (:468-469)
            for (uint32_t k : all_ks) {
                float pad_k = std::ceil(float(q) / float(k)) * float(k) - float(q);
                float pad_b = std::ceil(float(q) / float(best)) * float(best) - float(q);
                if (pad_k < pad_b) best = k;
            }
        }
        p.nk_sum[q] = {best};
    }
    return p;
}

std::vector<KmxPlanEntry> make_plan_entries(const std::vector<uint32_t>& ks, uint32_t range)
{
    Plan p = make_plan(ks, range);
    auto elem_of = [&](uint32_t k) -> uint8_t {
        for (size_t i = 0; i < ks.size(); ++i)
            if (ks[i] == k) return uint8_t(i);
        return 0;
    };
    std::vector<KmxPlanEntry> out(range);
    for (uint32_t q = 0; q < range; ++q) {
        KmxPlanEntry e{};
        const auto& sum = p.nk_sum[q];
        // kmer_index.hpp:512: the multi scheme is used only when the table says so
        // AND more than one k is instantiated.  Entries of the DP chain keep their
        // MULTI form even when they are a single summand, so that chains can be
        // walked; a 1-summand MULTI entry is served as an exact lookup (:529-530).
        if (p.use_multi[q] && ks.size() > 1) {
            e.scheme = KMX_SCHEME_MULTI;
            e.elem = elem_of(sum.back());
            e.nparts = uint16_t(sum.size());
        } else {
            e.scheme = KMX_SCHEME_SINGLE;
            e.elem = elem_of(sum.at(0));
            e.nparts = 1;
        }
        out[q] = e;
    }
    return out;
}

// The engine's own table for long queries.  The reference plans a query longer than every k on the k that wastes the fewest
// letters (ceil(q / k) * k - q, kmer_index.hpp:465-473) — for q = 13 on {8, 10, 12} that is k = 8, whose buckets are 256 times
// as long as those of k = 12 (two buckets of 1526 positions to intersect at 1e8 letters, against two of 6).  Which element
// answers does not change WHAT is answered — every occurrence of the query, ascending — so for single-k entries with q > k the
// device table names the LARGEST k <= q instead (shortest buckets), unless either choice could run into the sub-k fan-out guard
// through its rest (:119-122 via :234: the status must stay the reference's).  Sums of the multi-k scheme are re-planned the same
// way (KMX_SCHEME_REPLANNED, below); exact lengths and sub-k lengths are the reference's.  Searches that expose the reference's result object (KMX_SEARCH_KEEP_MASKS: candidate run
// + compressed_bitset of the element the REFERENCE would use) run on the reference's table.
std::vector<KmxPlanEntry> make_fast_plan_entries(const std::vector<uint32_t>& ks, uint32_t range, uint32_t sigma)
{
    std::vector<KmxPlanEntry> out = make_plan_entries(ks, range);
    if (ks.size() < 2) return out;
    auto fan_out = [&](uint32_t e) -> bool {             // sigma^e > KMX_SUBK_FANOUT_LIMIT ?
        unsigned __int128 v = 1;
        for (uint32_t t = 0; t < e; ++t) { v *= sigma; if (v > KMX_SUBK_FANOUT_LIMIT) return true; }
        return false;
    };
    for (uint32_t q = 1; q < range; ++q) {
        KmxPlanEntry& e = out[q];
        auto risky = [&](uint32_t k) { const uint32_t r = q % k; return r != 0 && fan_out(k - r); };
        if (e.scheme == KMX_SCHEME_MULTI && e.nparts >= 2) {
            // A sum of several ks (kmer_index.hpp:427-443) is also a run of parts of ONE k with the k-mer that ends the query as
            // the last of them: on the LARGEST k of the index every bucket is the shortest the index has (20 letters on {8, 10, 12}:
            // two 12-mers, 6 positions each at 1e8 letters, where the sum 10 + 10 intersects two buckets of 95).  The reference's
            // entry stays in place — longer sums walk through it (elem = its last summand, nparts = their number) — and the
            // element the engine uses instead rides in the upper bits of nparts.  Not when the rest could reach the fan-out guard
            // (the multi-k scheme never throws there: the status must stay the reference's), not for sums too long to encode.
            if (e.nparts > KMX_PLAN_NPARTS_MASK) continue;
            uint32_t best_k = 0, best_i = 0;
            for (size_t i = 0; i < ks.size(); ++i)
                if (ks[i] < q && ks[i] > best_k && !risky(ks[i])) { best_k = ks[i]; best_i = uint32_t(i); }
            if (best_k) {
                e.scheme = KMX_SCHEME_REPLANNED;
                e.nparts = uint16_t(e.nparts | (best_i << KMX_PLAN_ALT_SHIFT));
            }
            continue;
        }
        if (e.scheme != KMX_SCHEME_SINGLE) continue;
        const uint32_t k0 = ks[e.elem];
        if (q <= k0) continue;
        if (risky(k0)) continue;
        uint32_t best_k = k0;
        for (size_t i = 0; i < ks.size(); ++i)
            if (ks[i] <= q && ks[i] > best_k && !risky(ks[i])) { best_k = ks[i]; e.elem = uint8_t(i); }
    }
    return out;
}

// choose_best_k.hpp:12-60.  Candidates in descending priority (:22-23); every query length gives points to the
// FIRST candidate that divides it (3 points, :32-36) or misses a multiple by at most 3 (4 - miss points, :38-42);
// the candidates are then ordered by score (:50-51) and the first n_k returned (:55-57).  The reference sorts
// with std::sort (order of equal scores unspecified); equal scores keep the priority order here.
std::vector<uint32_t> choose_best_k(const uint64_t* lengths, uint64_t n_lengths, uint32_t n_k)
{
    static const uint32_t candidates[] = {29, 27, 25, 23, 21, 19, 17, 13, 11, 10};
    std::vector<std::pair<uint32_t, uint64_t>> k_and_score;
    for (uint32_t k : candidates) k_and_score.emplace_back(k, 0);
    for (uint64_t j = 0; j < n_lengths; ++j) {
        const uint64_t i = lengths[j];
        for (auto& p : k_and_score) {
            const uint64_t k = p.first;
            if (i % k == 0) { p.second += 3; break; }
            if (k - (i % k) <= 3) { p.second += 4 - (k - (i % k)); break; }
        }
    }
    std::stable_sort(k_and_score.begin(), k_and_score.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
    std::vector<uint32_t> out;
    for (uint32_t i = 0; i < n_k && i < k_and_score.size(); ++i) out.push_back(k_and_score[i].first);
    return out;
}

static uint32_t ceil_log2_u64(uint64_t v)
{
    uint32_t l = 0;
    while ((uint64_t(1) << l) < v) ++l;
    return l;
}

void build_slots(ElemImage& im)
{
    uint64_t u = im.ukeys.size();
    im.log2cap = std::max<uint32_t>(4, ceil_log2_u64(2 * std::max<uint64_t>(u, 1)));   // load <= 0.5
    uint64_t cap = uint64_t(1) << im.log2cap;
    im.slots.assign(cap, KmxSlot{0, 0, 0});
    for (uint64_t i = 0; i < u; ++i) {
        uint64_t s = slot_hash(im.ukeys[i], im.log2cap);
        while (im.slots[s].cnt != 0) s = (s + 1) & (cap - 1);   // linear probing
        // off addresses the line-aligned copy when the element has one (exact lookups / stitch candidates read it)
        im.slots[s] = KmxSlot{im.ukeys[i], im.aoffs.empty() ? im.offs[i] : im.aoffs[i], im.offs[i + 1] - im.offs[i]};
    }
}

uint32_t resolve_table_kind(uint32_t sigma, uint32_t k, uint64_t n, uint32_t requested)
{
    if (requested != KMX_TABLE_AUTO) return requested;
    const uint64_t n_keys = key_space(sigma, k), npos = n - k + 1, HIST_MAX = uint64_t(1) << 30;
    return (n_keys <= 4 * npos && n_keys <= HIST_MAX) ? KMX_TABLE_DENSE : KMX_TABLE_OPEN;
}

void add_aligned_copy(ElemImage& im)
{
    const uint64_t n_groups = im.offs.size() - 1;
    uint64_t present = 0;
    for (uint64_t g = 0; g < n_groups; ++g) present += im.offs[g + 1] != im.offs[g];
    im.region = im.npos;
    im.aoffs.clear();
    if (present == 0 || im.npos < 32 * present) return;          // short buckets: padding would cost more than it saves
    auto up32 = [](uint64_t v) { return (v + 31) & ~uint64_t(31); };
    uint64_t cur = up32(im.npos);
    uint64_t end = cur;
    for (uint64_t g = 0; g < n_groups; ++g) end += up32(im.offs[g + 1] - im.offs[g]);
    if (end >= 0xFFFFFFFFull) return;
    im.positions.resize(end, 0);
    im.aoffs.resize(n_groups);
    for (uint64_t g = 0; g < n_groups; ++g) {
        const uint32_t a = im.offs[g], c = im.offs[g + 1] - a;
        im.aoffs[g] = uint32_t(cur);
        std::copy(im.positions.begin() + a, im.positions.begin() + a + c, im.positions.begin() + cur);
        cur += up32(c);
    }
    im.region = end;
    if (im.table_kind == KMX_TABLE_DENSE) {
        im.atab.resize(n_groups + 1);
        for (uint64_t g = 0; g < n_groups; ++g) im.atab[g] = ((im.aoffs[g] >> 5) << 5) | ((im.offs[g + 1] - im.offs[g]) & 31u);
        im.atab[n_groups] = uint32_t(end >> 5) << 5;
        im.aoffs.clear();
    }
}

bool flatten_element(const uint8_t* ranks, uint64_t n, uint32_t sigma, uint32_t k, uint32_t table_kind,
                     ElemImage& im, std::string& err, bool aligned_copy)
{
    if (!k_is_valid(sigma, k)) { err = "k must satisfy 0 < k < 64 / log2(sigma)"; return false; }
    if (n < k) { err = "text shorter than k"; return false; }
    if (n + k - 1 >= 0xFFFFFFFFull) { err = "your text is too large for this configuration"; return false; }   // :169-170
    im = ElemImage();
    im.k = k;
    im.npos = n - k + 1;
    im.n_keys = key_space(sigma, k);
    const uint64_t HIST_MAX = uint64_t(1) << 30;
    if (table_kind == KMX_TABLE_AUTO)
        table_kind = (im.n_keys <= 4 * im.npos && im.n_keys <= HIST_MAX) ? KMX_TABLE_DENSE : KMX_TABLE_OPEN;
    if (table_kind == KMX_TABLE_DENSE && im.n_keys > HIST_MAX) {
        err = "dense table requested but sigma^k exceeds 2^30 keys";
        return false;
    }
    im.table_kind = table_kind;
    im.positions.resize(im.npos);
    const uint64_t top = key_space(sigma, k - 1);

    auto first_hash = [&]() {
        uint64_t h = 0;
        for (uint32_t i = 0; i < k; ++i) h = h * sigma + ranks[i];   // == sum r_i * sigma^(k-i-1), kmer_index.hpp:56-73
        return h;
    };

    if (im.n_keys <= HIST_MAX) {
        // counting sort by hash; stable, so positions ascend inside a group
        std::vector<uint32_t> start(im.n_keys + 1, 0);
        uint64_t h = first_hash();
        for (uint64_t i = 0;; ++i) {
            ++start[h + 1];
            if (i + 1 >= im.npos) break;
            h = (h - uint64_t(ranks[i]) * top) * sigma + ranks[i + k];
        }
        for (uint64_t j = 0; j < im.n_keys; ++j) start[j + 1] += start[j];
        std::vector<uint32_t> cursor(start.begin(), start.end() - 1);
        h = first_hash();
        for (uint64_t i = 0;; ++i) {
            im.positions[cursor[h]++] = uint32_t(i);
            if (i + 1 >= im.npos) break;
            h = (h - uint64_t(ranks[i]) * top) * sigma + ranks[i + k];
        }
        im.region = im.npos;
        if (table_kind == KMX_TABLE_DENSE) {
            im.offs.swap(start);
            if (aligned_copy) add_aligned_copy(im);
        } else {
            im.offs.push_back(0);
            for (uint64_t j = 0; j < im.n_keys; ++j)
                if (start[j + 1] != start[j]) { im.ukeys.push_back(j); im.offs.push_back(start[j + 1]); }
            if (aligned_copy) add_aligned_copy(im);
            build_slots(im);
        }
    } else {
        // key space too large for a histogram: sort (hash, position) pairs
        std::vector<std::pair<uint64_t, uint32_t>> pairs(im.npos);
        uint64_t h = first_hash();
        for (uint64_t i = 0;; ++i) {
            pairs[i] = {h, uint32_t(i)};
            if (i + 1 >= im.npos) break;
            h = (h - uint64_t(ranks[i]) * top) * sigma + ranks[i + k];
        }
        std::sort(pairs.begin(), pairs.end());
        im.offs.push_back(0);
        for (uint64_t i = 0; i < im.npos; ++i) {
            im.positions[i] = pairs[i].second;
            if (i + 1 == im.npos || pairs[i + 1].first != pairs[i].first) {
                im.ukeys.push_back(pairs[i].first);
                im.offs.push_back(uint32_t(i + 1));
            }
        }
        im.region = im.npos;
        if (aligned_copy) add_aligned_copy(im);
        build_slots(im);
    }
    return true;
}

} // namespace kmx
