// mt_block_test.cpp -- Mt19937Block against std::mt19937: seeds (0, 1, the default, cfg3's 666666, all ones) and two
// million words each, across many refills.  Prints "ok" and the rate of both engines.
#include <chrono>
#include <cstdio>
#include <random>

#include "mt_block.h"

int main() {
  const unsigned seeds[] = {0u, 1u, 5489u, 666666u, 0xffffffffu, 2463534242u};
  for (unsigned seed : seeds) {
    std::mt19937 ref(seed);
    slamhip::Mt19937Block got(seed);
    for (int i = 0; i < 2000000; ++i) {
      const unsigned a = ref(), b = got();
      if (a != b) {
        std::printf("mismatch: seed %u word %d: %u != %u\n", seed, i, a, b);
        return 1;
      }
    }
  }
  unsigned acc = 0;
  std::mt19937 ref(7);
  slamhip::Mt19937Block blk(7);
  const int n = 20000000;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) acc ^= ref();
  const auto t1 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) acc ^= blk();
  const auto t2 = std::chrono::steady_clock::now();
  std::printf("ok %u std %.2f ns/word block %.2f ns/word\n", acc, 1e9 * std::chrono::duration<double>(t1 - t0).count() / n,
              1e9 * std::chrono::duration<double>(t2 - t1).count() / n);
  return 0;
}
