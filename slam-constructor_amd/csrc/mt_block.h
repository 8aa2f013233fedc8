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

private:
  void refill();
  uint32_t s_[624], out_[624];
  int p_;
};

}  // namespace slamhip
