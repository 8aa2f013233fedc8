#include <chrono>
#include <cstdio>
#include "matchers.h"
int main() {
  using namespace slamhip;
  PairTape tape(666666u);
  const size_t per_match = 6146;
  auto t0 = std::chrono::steady_clock::now();
  size_t upto = 0;
  double s = 0;
  for (int m = 0; m < 200; ++m) {
    upto += per_match;
    tape.prefetch(upto, 1 << 30);
    s += tape.at(upto - 1).ret;
  }
  auto t1 = std::chrono::steady_clock::now();
  std::printf("%.1f us per match's tape (%zu pairs)  %g\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 200, per_match, s);
}
