// mt_block.cpp -- std::mt19937, bit for bit, 624 tempered words per refill (plain C++, no device code).
//
// The Monte-Carlo matcher's candidates come from three std::normal_distributions over ONE std::mt19937
// (src/core/scan_matchers/monte_carlo_scan_matcher.h:26, random_utils.h:17-34); a cfg3 match draws ~6200 Marsaglia
// pairs = ~31 000 engine words, and on the device chain the host's only work per match is producing them.  libstdc++'s
// engine twists its state with a scalar loop (6.9 ns per word measured here, 2.4 with AVX2 code generation): this one
// keeps the recurrence of [rand.eng.mers] (w 32, n 624, m 397, r 31, a 0x9908b0df, u 11, s 7, b 0x9d2c5680, t 15,
// c 0xefc60000, l 18) but writes the new state into a second array in three stretches none of which reads what it
// writes, so that the compiler vectorizes them, and is compiled twice -- baseline x86-64 and AVX2, picked at run time.
// tests/native/mt_block_test.cpp checks seeds and long runs against std::mt19937.
#include <cstdint>

#include "mt_block.h"

namespace slamhip {
namespace {

constexpr uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMag = 0x9908b0dfu;

static inline __attribute__((always_inline)) uint32_t twist(uint32_t hi, uint32_t lo, uint32_t far) {
  const uint32_t y = (hi & kUpper) | (lo & kLower);
  return far ^ (y >> 1) ^ ((0u - (y & 1u)) & kMag);
}

static inline __attribute__((always_inline)) void refill_body(uint32_t *__restrict s, uint32_t *__restrict out) {
  uint32_t t[624];
  // x_k = x_{k+397 mod n} ^ twist(x_k, x_{k+1}): k < 227 reads only old words; 227 <= k < 454 reads the new words
  // 0..226; 454 <= k < 623 the new words 227..395; the last one wraps around to the new word 0
  for (int k = 0; k < 227; ++k) t[k] = twist(s[k], s[k + 1], s[k + 397]);
  for (int k = 227; k < 454; ++k) t[k] = twist(s[k], s[k + 1], t[k - 227]);
  for (int k = 454; k < 623; ++k) t[k] = twist(s[k], s[k + 1], t[k - 227]);
  t[623] = twist(s[623], t[0], t[396]);
  for (int k = 0; k < 624; ++k) {
    uint32_t z = t[k];
    s[k] = z;
    z ^= z >> 11;
    z ^= (z << 7) & 0x9d2c5680u;
    z ^= (z << 15) & 0xefc60000u;
    z ^= z >> 18;
    out[k] = z;
  }
}

static inline __attribute__((always_inline)) double canonical2(uint32_t lo, uint32_t hi) {
  const double ret = ((double)lo + (double)hi * 4294967296.0) * 0x1p-64;  // (sum / 2^64: a power of two, exact)
  return ret >= 1.0 ? 0x1.fffffffffffffp-1 : ret;                         // std::nextafter(1.0, 0.0)
}

static inline __attribute__((always_inline)) void attempts_body(const uint32_t *__restrict w, int n, double *__restrict x,
                                                                double *__restrict y, double *__restrict r2) {
  for (int a = 0; a < n; ++a) {
    const double xx = 2.0 * canonical2(w[4 * a], w[4 * a + 1]) - 1.0;
    const double yy = 2.0 * canonical2(w[4 * a + 2], w[4 * a + 3]) - 1.0;
    x[a] = xx;
    y[a] = yy;
    r2[a] = xx * xx + yy * yy;
  }
}

void attempts_generic(const uint32_t *w, int n, double *x, double *y, double *r2) { attempts_body(w, n, x, y, r2); }
__attribute__((target("avx2"))) void attempts_avx2(const uint32_t *w, int n, double *x, double *y, double *r2) {
  attempts_body(w, n, x, y, r2);
}

void refill_generic(uint32_t *s, uint32_t *out) { refill_body(s, out); }
__attribute__((target("avx2"))) void refill_avx2(uint32_t *s, uint32_t *out) { refill_body(s, out); }

}  // namespace

Mt19937Block::Mt19937Block(uint32_t seed) {
  s_[0] = seed;
  for (uint32_t i = 1; i < 624; ++i) s_[i] = 1812433253u * (s_[i - 1] ^ (s_[i - 1] >> 30)) + i;
  p_ = 624;
}

void polar_attempts(const uint32_t *words, int n, double *x, double *y, double *r2) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) attempts_avx2(words, n, x, y, r2);
  else attempts_generic(words, n, x, y, r2);
}

void Mt19937Block::refill() {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) refill_avx2(s_, out_);
  else refill_generic(s_, out_);
  p_ = 0;
}

}  // namespace slamhip
