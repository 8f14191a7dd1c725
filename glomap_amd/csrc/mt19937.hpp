// mt19937.hpp — the reference's random start, drawn fast and bit for bit.
//
// GlobalPositioner draws its start with `RandVector3d(random_generator_, -1, 1)` (global_positioning.cc:135,261,449): a
// std::mt19937 seeded with options.seed feeding std::uniform_real_distribution<double>(-1, 1).  The draws decide the whole
// LM trajectory, so parity needs exactly that stream — but libstdc++'s scalar engine is 12.7 ms of single-threaded host
// time for the 3 M doubles of configs[3], as much as 50 PCG iterations.  This is the same generator with the state
// refreshed a block (624 words) at a time in loops the compiler vectorises (the recurrence reaches back 227 words, far
// more than a vector), and the two library layers above it restated:
//   std::generate_canonical<double, 53>(g):  k = 2 calls,  sum = g() + g() * 2^32  (both terms exact in double, the sum rounded
//       to nearest-even — the same as converting the 64-bit integer),  u = sum / 2^64,  u >= 1 -> nextafter(1, 0);
//   std::uniform_real_distribution<double>(a, b)(g) = u * (b - a) + a.
// tests/test_abi.py pins it against std::mt19937 + std::uniform_real_distribution through gsfm_selftest_mt19937.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>

namespace gsfm {

class FastMt19937 {
 public:
  explicit FastMt19937(uint32_t seed = 5489u) { this->seed(seed); }
  void seed(uint32_t s) {
    mt_[0] = s;
    for (uint32_t i = 1; i < kN; ++i) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + i;
    idx_ = kN;
  }
  // next 32-bit output of std::mt19937
  uint32_t next() {
    if (idx_ >= kN) refill();
    return out_[idx_++];
  }
  // skip n outputs (std::mt19937::discard)
  void discard(unsigned long long n) {
    while (n > 0) {
      if (idx_ >= kN) refill();
      const unsigned long long take = n < (unsigned long long)(kN - idx_) ? n : (unsigned long long)(kN - idx_);
      idx_ += (int)take;
      n -= take;
    }
  }
  // scale * std::uniform_real_distribution<double>(-1, 1)(g), `count` of them
  void fill_uniform_pm1(double* dst, size_t count, double scale) {
    size_t done = 0;
    while (done < count) {
      if (idx_ >= kN) refill();
      if (((kN - idx_) & 1) != 0 || idx_ + 1 >= kN) {  // odd phase or a pair straddling two blocks: scalar
        const uint32_t lo = next(), hi = next();
        dst[done++] = scale * to_pm1(lo, hi);
        continue;
      }
      const size_t pairs = (size_t)(kN - idx_) / 2;
      const size_t take = pairs < count - done ? pairs : count - done;
      const uint32_t* o = out_ + idx_;
      for (size_t i = 0; i < take; ++i) dst[done + i] = scale * to_pm1(o[2 * i], o[2 * i + 1]);
      idx_ += (int)(2 * take);
      done += take;
    }
  }
  double uniform_pm1() {
    const uint32_t lo = next(), hi = next();
    return to_pm1(lo, hi);
  }

 private:
  static constexpr int kN = 624, kM = 397;
  static double to_pm1(uint32_t lo, uint32_t hi) {
    double u = (double)(((uint64_t)hi << 32) | lo) * 0x1p-64;  // generate_canonical<double, 53>
    if (u >= 1.0) u = std::nextafter(1.0, 0.0);
    return u * 2.0 + (-1.0);  // uniform_real_distribution(-1, 1): u * (b - a) + a
  }
  void refill() {
    constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MAT = 0x9908b0dfu;
    uint32_t* __restrict__ mt = mt_;
    for (int i = 0; i < kN - kM; ++i) {  // reads mt[i + 397]: old words
      const uint32_t y = (mt[i] & UP) | (mt[i + 1] & LO);
      mt[i] = mt[i + kM] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);
    }
    for (int i = kN - kM; i < kN - 1; ++i) {  // reads mt[i - 227]: words refreshed 227 steps ago
      const uint32_t y = (mt[i] & UP) | (mt[i + 1] & LO);
      mt[i] = mt[i - (kN - kM)] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);
    }
    {
      const uint32_t y = (mt[kN - 1] & UP) | (mt[0] & LO);
      mt[kN - 1] = mt[kM - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);
    }
    uint32_t* __restrict__ o = out_;
    for (int i = 0; i < kN; ++i) {  // tempering
      uint32_t y = mt[i];
      y ^= y >> 11;
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= y >> 18;
      o[i] = y;
    }
    idx_ = 0;
  }
  uint32_t mt_[kN];
  uint32_t out_[kN];
  int idx_ = kN;
};

}  // namespace gsfm
