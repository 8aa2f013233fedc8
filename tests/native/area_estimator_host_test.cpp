// tests/native/area_estimator_host_test.cpp -- the DEVICE restatement of AreaOccupancyEstimator::estimate_occupancy
// (slam-constructor_amd/csrc/area_estimator_device.h: edge slots with flags, everything in registers) compiled for
// the host and run against the oracle's vector-shaped restatement (oracle/area_estimator.h, pinned to the reference's
// own 64 unit-test cases and to goldens of the compiled reference) on beams chosen to hit the special cases:
// end points on cell corners and edges, beams along edge lines, beams through two corners, cells far from the
// origin, plus a random bulk.  Bitwise equality of (prob, qual), NaN included.
//   g++ -std=c++17 -O2 -ffp-contract=off -I<repo> area_estimator_host_test.cpp
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#define __device__
#define __forceinline__ inline
#include "slam-constructor_amd/csrc/area_estimator_device.h"

namespace orc {
#include "oracle/area_estimator.h"
}

static bool same_bits(double a, double b) {
  if (std::isnan(a) && std::isnan(b)) return true;
  return std::memcmp(&a, &b, sizeof a) == 0;
}

int main() {
  std::mt19937_64 eng(12345);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  const double base4[4] = {0.95, 1.0, 0.01, 1.0};
  long long n = 0, bad = 0, invalid = 0, occ_hits = 0;
  const double scales[3] = {0.025, 0.05, 0.1};
  for (int it = 0; it < 3000000; ++it) {
    const double scale = scales[it % 3];
    const int cx = (int)(U(eng) * 40) - 20 + ((it & 7) == 0 ? 4000 : 0), cy = (int)(U(eng) * 40) - 20;
    const double bot = scale * cy, top = scale * (cy + 1), left = scale * cx, right = scale * (cx + 1);
    auto pick = [&](double lo, double hi) {
      // a coordinate that is often special: an edge, the middle, an edge of a neighbouring cell, a hair off an edge
      const int kind = (int)(U(eng) * 10);
      switch (kind) {
        case 0: return lo;
        case 1: return hi;
        case 2: return 0.5 * (lo + hi);
        case 3: return lo - (hi - lo);
        case 4: return hi + (hi - lo);
        case 5: return lo + 1e-9;
        case 6: return std::nextafter(hi, 1e9);
        default: return lo - 3 * (hi - lo) + U(eng) * 7 * (hi - lo);
      }
    };
    const double bx = pick(left, right), by = pick(bot, top), ex = pick(left, right), ey = pick(bot, top);
    const int is_occ = (it >> 3) & 1;
    const double shift = 0.01 * scale;
    const slamhip::ae::ae_occ d = slamhip::ae::ae_estimate(slamhip::ae::ae_pt{bx, by}, slamhip::ae::ae_pt{ex, ey},
                                                           slamhip::ae::ae_rect{bot, top, left, right}, is_occ, base4, shift);
    const orc::ae_occ o = orc::ae_estimate(orc::ae_pt{bx, by}, orc::ae_pt{ex, ey}, orc::ae_rect{bot, top, left, right},
                                           is_occ, base4, shift);
    ++n;
    invalid += std::isnan(d.prob);
    occ_hits += is_occ && !std::isnan(d.prob) && d.prob > 0.011;
    if (!same_bits(d.prob, o.prob) || !same_bits(d.qual, o.qual)) {
      if (bad < 5)
        std::printf("MISMATCH beam (%.17g, %.17g)-(%.17g, %.17g) cell [%g %g %g %g] occ %d: device (%.17g, %.17g) oracle "
                    "(%.17g, %.17g)\n", bx, by, ex, ey, bot, top, left, right, is_occ, d.prob, d.qual, o.prob, o.qual);
      ++bad;
    }
  }
  std::printf("%lld cases, %lld invalid, %lld occupied estimates above base_empty, %lld mismatches\n", n, invalid, occ_hits, bad);
  return (bad == 0 && invalid > n / 50 && invalid < n - n / 50 && occ_hits > n / 100) ? 0 : 1;
}
