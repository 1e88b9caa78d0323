// Minimal alphabet layer for the host mirror.
//
// The reference is templated on seqan3 alphabets and consumes exactly three things from them:
// seqan3::alphabet_size<T>, seqan3::to_rank(letter) (kmer_index.hpp:50,59,128) and, in its
// generators, T{}.assign_rank(r) (benchmarks/input_generator.hpp:60).  seqan3 is not part of
// this repository, so the same surface is provided here by small rank-wrapper types; any other
// type with `static constexpr alphabet_size` and `to_rank()` (or a real seqan3 alphabet, when
// seqan3 is on the include path) plugs into kmer::kmer_index through alphabet_traits.
//
// Rank tables follow seqan3 (dna4 ACGT = 0..3, dna5 ACGNT = 0..4, aa20 alphabetical, ...); they
// only matter for char I/O — the engine sees ranks.  No reference test pins them.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string_view>
#include <type_traits>

#if __has_include(<seqan3/alphabet/concept.hpp>)
#include <seqan3/alphabet/concept.hpp>
#define KMX_HAVE_SEQAN3 1
#endif

namespace kmer::alphabet
{
    template<std::size_t sigma, const char* chars>
    struct rank_alphabet
    {
        static constexpr std::size_t alphabet_size = sigma;
        std::uint8_t rank = 0;

        constexpr std::uint8_t to_rank() const noexcept { return rank; }
        constexpr rank_alphabet& assign_rank(std::uint8_t r) noexcept { rank = r; return *this; }
        constexpr char to_char() const noexcept { return chars[rank]; }
        constexpr rank_alphabet& assign_char(char c) noexcept
        {
            rank = 0;
            for (std::size_t i = 0; i < sigma; ++i)
                if (chars[i] == c || chars[i] == (c >= 'a' && c <= 'z' ? c - 32 : c)) { rank = std::uint8_t(i); break; }
            return *this;
        }
        friend constexpr bool operator==(rank_alphabet a, rank_alphabet b) noexcept { return a.rank == b.rank; }
        friend constexpr bool operator!=(rank_alphabet a, rank_alphabet b) noexcept { return a.rank != b.rank; }
    };

    inline constexpr char dna4_chars[] = "ACGT";
    inline constexpr char dna5_chars[] = "ACGNT";
    inline constexpr char dna15_chars[] = "ABCDGHKMNRSTVWY";
    inline constexpr char aa20_chars[] = "ACDEFGHIKLMNPQRSTVWY";
    inline constexpr char aa27_chars[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZ*";

    using dna4 = rank_alphabet<4, dna4_chars>;
    using dna5 = rank_alphabet<5, dna5_chars>;
    using dna15 = rank_alphabet<15, dna15_chars>;
    using aa20 = rank_alphabet<20, aa20_chars>;
    using aa27 = rank_alphabet<27, aa27_chars>;
} // namespace kmer::alphabet

namespace kmer::detail
{
    template<typename T, typename = void>
    struct alphabet_traits
    {
#ifdef KMX_HAVE_SEQAN3
        static constexpr std::size_t size = seqan3::alphabet_size<T>;
        static std::uint8_t to_rank(T const& l) { return std::uint8_t(seqan3::to_rank(l)); }
#else
        static_assert(sizeof(T) == 0, "alphabet type needs `static constexpr alphabet_size` and `to_rank()`");
#endif
    };

    template<typename T>
    struct alphabet_traits<T, std::void_t<decltype(T::alphabet_size), decltype(std::declval<T const&>().to_rank())>>
    {
        static constexpr std::size_t size = T::alphabet_size;
        static std::uint8_t to_rank(T const& l) { return std::uint8_t(l.to_rank()); }
    };

    // k > 0 and k < 64 / log2(sigma)  (static_assert of kmer_index.hpp:42-43)  <=>  sigma^k < 2^64
    constexpr bool k_is_valid(std::size_t sigma, std::size_t k)
    {
        if (k == 0 || sigma < 2) return false;
        unsigned __int128 v = 1;
        for (std::size_t i = 0; i < k; ++i)
        {
            v *= sigma;
            if (v >> 64) return false;
        }
        return true;
    }
} // namespace kmer::detail
