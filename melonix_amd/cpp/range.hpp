// range.hpp — the three names the melonix UI and the facade share (reference range.hpp:4-18):
//   MaxRanges  capacity of the column caches
//   Range      one spectrogram column as a (start, end) sample-index pair
//   pair_hash  hasher so that Range can key unordered containers
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <utility>

inline constexpr int MaxRanges = 4000;

using Range = std::pair<int, int>;

struct pair_hash {
  // 64-bit mix of the two member hashes (splitmix64 finaliser); any decent mix works here,
  // callers only rely on equal pairs hashing equally.
  template <class A, class B>
  std::size_t operator()(const std::pair<A, B> &p) const noexcept {
    std::uint64_t x = static_cast<std::uint64_t>(std::hash<A>{}(p.first)) * 0x9e3779b97f4a7c15ull +
                      static_cast<std::uint64_t>(std::hash<B>{}(p.second));
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return static_cast<std::size_t>(x);
  }
};
