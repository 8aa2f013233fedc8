// pair_tape_test.cpp -- the Monte-Carlo matcher's pair tape (matchers.h PairTape: block engine, vectorized polar
// attempts) against the real thing: std::normal_distribution<double> over std::mt19937, which hands out a pair's two
// values on consecutive calls.  Four seeds, 300 000 pairs each, bit for bit; then three distributions sharing one
// engine the way GaussianPoseEnumerator's do (random_utils.h:17-34), drawn through TapeNormal.
#include <cstdio>
#include <cstring>
#include <random>

#include "matchers.h"

int main() {
  using namespace slamhip;
  const unsigned seeds[] = {666666u, 0u, 1u, 0xfffffffeu};
  for (unsigned seed : seeds) {
    PairTape tape(seed);
    std::mt19937 e(seed);
    std::normal_distribution<double> nd(0.0, 1.0);
    for (size_t i = 0; i < 300000; ++i) {
      const PairTape::Pair p = tape.at(i);
      const double a = nd(e), b = nd(e);
      if (std::memcmp(&a, &p.ret, 8) != 0 || std::memcmp(&b, &p.saved, 8) != 0) {
        std::printf("mismatch: seed %u pair %zu\n", seed, i);
        return 1;
      }
    }
  }
  {
    PairTape tape(666666u);
    std::mt19937 e(666666u);
    std::normal_distribution<double> nx(0.0, 0.2), ny(0.0, 0.2), nt(0.0, 0.1);
    TapeNormal tx(0.0, 0.2), ty(0.0, 0.2), tt(0.0, 0.1);
    size_t pos = 0;
    for (int i = 0; i < 200000; ++i) {
      const double a[3] = {nx(e), ny(e), nt(e)};
      const double b[3] = {tx.draw(tape, pos), ty.draw(tape, pos), tt.draw(tape, pos)};
      if (std::memcmp(a, b, sizeof(a)) != 0) {
        std::printf("mismatch: candidate %d\n", i);
        return 1;
      }
    }
  }
  std::printf("ok\n");
  return 0;
}
