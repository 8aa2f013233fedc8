// mt_block.h -- std::mt19937's output sequence from a block generator (mt_block.cpp)
#pragma once
#include <cstdint>

namespace slamhip {

class Mt19937Block {
public:
  using result_type = uint32_t;
  explicit Mt19937Block(uint32_t seed = 5489u);
  uint32_t operator()() {
    if (p_ >= 624) refill();
    return out_[p_++];
  }
  static constexpr uint32_t min() { return 0u; }
  static constexpr uint32_t max() { return 0xffffffffu; }
  // the next 624 words at once (only while the engine stands at a block boundary: a consumer that takes its words
  // four at a time always does)
  const uint32_t *next_block() {
    refill();
    p_ = 624;
    return out_;
  }
  bool at_block_boundary() const { return p_ >= 624; }

private:
  void refill();
  uint32_t s_[624], out_[624];
  int p_;
};

// The first half of std::normal_distribution's Marsaglia polar method for `n` attempts of four engine words each:
// x = 2 c(w0, w1) - 1, y = 2 c(w2, w3) - 1, r2 = x x + y y with c = std::generate_canonical<double, 53> over a 32-bit
// engine (low word first, the sum rounded once, a result of 1 replaced by its predecessor).  Vectorized (AVX2 clone).
void polar_attempts(const uint32_t *words, int n, double *x, double *y, double *r2);

}  // namespace slamhip
