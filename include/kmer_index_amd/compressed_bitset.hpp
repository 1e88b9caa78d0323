// kmer::detail::compressed_bitset — host mirror of the reference class of the same name
// (compressed_bitset.hpp:9-105): same public members, same word layout (n_bits/64 + 1 words,
// bit i = word i>>6, bit i&63, LSB first), same std::out_of_range behaviour.  Additionally it
// can be built from the mask words the GPU produced (k_validate's ballots).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace kmer::detail
{
    template<typename integer_t = std::uint_fast64_t>
    class compressed_bitset
    {
        static_assert(sizeof(integer_t) == 8, "mask words are 64 bit");
        std::size_t _n_bits;
        std::vector<integer_t> _bits;

        void check(std::size_t i) const
        {
            if (i >= _n_bits) throw std::out_of_range("compressed bitset index out of range");
        }

    public:
        compressed_bitset(std::size_t n_bits, bool zero_or_one)
            : _n_bits(n_bits), _bits(std::max<std::size_t>(n_bits / 64 + 1, 1), zero_or_one ? ~integer_t(0) : integer_t(0))
        {}

        // adopt device-produced words (n_bits/64 + 1 of them)
        compressed_bitset(std::size_t n_bits, const std::uint64_t* words)
            : _n_bits(n_bits), _bits(words, words + (n_bits / 64 + 1))
        {}

        [[nodiscard]] std::vector<bool> to_vector() const
        {
            std::vector<bool> out;
            out.reserve(_n_bits);
            for (std::size_t i = 0; i < _n_bits; ++i) out.push_back(at(i));
            return out;
        }
        explicit operator std::vector<bool>() const { return to_vector(); }

        void set_0(std::size_t i) { check(i); _bits[i >> 6] &= ~(integer_t(1) << (i & 63)); }
        void set_1(std::size_t i) { check(i); _bits[i >> 6] |= integer_t(1) << (i & 63); }
        bool at(std::size_t i) const { check(i); return (_bits[i >> 6] >> (i & 63)) & 1; }
        void clear_to_1() { std::fill(_bits.begin(), _bits.end(), ~integer_t(0)); }
        void clear_to_0() { std::fill(_bits.begin(), _bits.end(), integer_t(0)); }
        std::size_t size() const { return _n_bits; }

        std::size_t count_bits_equal_to(bool b) const
        {
            std::size_t ones = 0;
            for (std::size_t w = 0; w * 64 < _n_bits; ++w)
            {
                integer_t word = _bits[w];
                std::size_t rem = _n_bits - w * 64;
                if (rem < 64) word &= (integer_t(1) << rem) - 1;
                ones += std::size_t(__builtin_popcountll(word));
            }
            return b ? ones : _n_bits - ones;
        }

        const std::vector<integer_t>& words() const { return _bits; }
    };
} // namespace kmer::detail
