// tests/native/libm_exact_check.cpp -- csrc/libm_exact.h against the RUNNING libm, bit for bit (CPU suite).
//   libm_exact_check <fma:0|1> <millions of arguments per function and range> [threads] [via_sincos:0|1]
// Arguments: uniformly random in the ranges the functions branch on (sin / cos: every interval of s_sin.c up to
// 105414350, the interval borders +- a few ulps, the 1/128 table knots, multiples of pi/2; exp: the GMapping range
// [-40, 0], the whole finite range, the special-case borders), drawn from splitmix64 with a fixed seed per thread.
// Prints the number of mismatches per function (0 expected) and the first few offenders; exit code 1 on any.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../slam-constructor_amd/csrc/libm_exact.h"

using namespace slamhip::libm_exact;

// libm's entry points through volatile pointers: sin(x) and cos(x) of one argument must stay two calls (gcc merges them
// into sincos(), which in glibc 2.35 is NOT the same code: sin / cos are picked per CPU (the FMA build where AVX2 + FMA
// are usable), sincos has no such variant and is always the plain build)
static double (*volatile p_sin)(double) = sin;
static double (*volatile p_cos)(double) = cos;
static double (*volatile p_exp)(double) = exp;
static void (*volatile p_sincos)(double, double *, double *) = sincos;
static int g_via_sincos = 0;  // 1: compare sin_ / cos_ with sincos()'s two results
static inline double ref_sin(double x) {
  if (!g_via_sincos || fabs(x) >= 0x1.921fbp+26) return p_sin(x);  // (>= 105414350: not restated, see libm_exact.h)
  double s, c;
  p_sincos(x, &s, &c);
  return s;
}
static inline double ref_cos(double x) {
  if (!g_via_sincos || fabs(x) >= 0x1.921fbp+26) return p_cos(x);
  double s, c;
  p_sincos(x, &s, &c);
  return c;
}

static inline uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static inline double u01(uint64_t &s) { return (double)(splitmix(s) >> 11) * 0x1p-53; }
static inline bool same(double a, double b) { return bits_of(a) == bits_of(b) || (a != a && b != b); }

struct Range { double lo, hi; };
static const Range kTrig[] = {{0.0, 0x1p-20}, {0x1p-28, 0x1p-24}, {0.0, 0.13}, {0.12, 0.86}, {0.85, 2.43}, {2.4, 7.0},
                              {0.0, 6.5}, {6.0, 1000.0}, {1000.0, 105414350.0}, {105414300.0, 105414350.0}};
static const Range kExp[] = {{-1.0, 0.0}, {-40.0, 0.0}, {-0x1p-50, 0x1p-50}, {-745.5, -700.0}, {-1100.0, -500.0}, {-710.0, 710.0},
                             {500.0, 1030.0}};

template <bool FMA>
static void worker(int tid, long per_range, std::atomic<long> *bad, double *first_bad) {
  uint64_t s = 0x1234567ull + 977ull * (uint64_t)tid;
  auto note = [&](int f, double x) {
    const long at = bad[f].fetch_add(1);
    if (at < 4) first_bad[4 * f + at] = x;
  };
  for (const Range &r : kTrig)
    for (long i = 0; i < per_range; ++i) {
      double x = r.lo + (r.hi - r.lo) * u01(s);
      if (splitmix(s) & 1) x = -x;
      if (!same(sin_<FMA>(x), ref_sin(x))) note(0, x);
      if (!same(cos_<FMA>(x), ref_cos(x))) note(1, x);
    }
  // structured: table knots k/128 +- a few ulps, interval borders, multiples of pi/2 and their neighbours
  for (long i = 0; i < per_range; ++i) {
    const uint64_t z = splitmix(s);
    double x;
    switch (z & 3) {
      case 0: x = (double)((z >> 8) % 900) / 128.0; break;
      case 1: { static const double b[] = {0x1p-26, 0x1p-27, 0.126, 0.855469, 0.85546875, 2.426265, 105414350.0}; x = b[(z >> 8) % 7]; } break;
      case 2: x = (double)((z >> 8) % 4000) * 1.5707963267948966; break;
      default: x = (double)((z >> 8) % 200000) * (0.5 / 128.0); break;
    }
    const long d = (long)((z >> 40) % 41) - 20;
    x = double_of(bits_of(x) + (uint64_t)d);
    if (z & 4) x = -x;
    if (!same(sin_<FMA>(x), ref_sin(x))) note(0, x);
    if (!same(cos_<FMA>(x), ref_cos(x))) note(1, x);
  }
  for (const Range &r : kExp)
    for (long i = 0; i < per_range; ++i) {
      const double x = r.lo + (r.hi - r.lo) * u01(s);
      if (!same(exp_<FMA>(x), p_exp(x))) note(2, x);
    }
}

int main(int argc, char **argv) {
  const int fma_variant = argc > 1 ? atoi(argv[1]) : 1;
  const double millions = argc > 2 ? atof(argv[2]) : 1.0;
  int threads = argc > 3 ? atoi(argv[3]) : (int)std::thread::hardware_concurrency();
  g_via_sincos = argc > 4 ? atoi(argv[4]) : 0;
  if (threads < 1) threads = 1;
  const long per_range = (long)(millions * 1e6 / threads) + 1;
  std::atomic<long> bad[3];
  for (auto &b : bad) b = 0;
  double first_bad[12];
  memset(first_bad, 0, sizeof first_bad);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back(fma_variant ? worker<true> : worker<false>, t, per_range, bad, first_bad);
  for (auto &t : pool) t.join();
  // edge values, once
  static const double edge[] = {0.0, -0.0, INFINITY, -INFINITY, NAN, 0x1p-1074, -0x1p-1074, 0x1p-1022, 709.782712893384, 709.782712893385,
                                -708.3964185322641, -745.1332191019411, -745.1332191019412, -1023.9, -1024.0, 1024.0, 512.0, -512.0};
  for (double x : edge) {
    const double want = p_exp(x), got = fma_variant ? exp_<true>(x) : exp_<false>(x);
    if (!same(got, want)) { if (bad[2].fetch_add(1) < 4) first_bad[8] = x; }
    if (fabs(x) < 105414350.0) {
      if (!same(fma_variant ? sin_<true>(x) : sin_<false>(x), ref_sin(x))) { if (bad[0].fetch_add(1) < 4) first_bad[0] = x; }
      if (!same(fma_variant ? cos_<true>(x) : cos_<false>(x), ref_cos(x))) { if (bad[1].fetch_add(1) < 4) first_bad[4] = x; }
    }
  }
  const long n_trig = per_range * threads * (long)(sizeof kTrig / sizeof kTrig[0] + 1);
  const long n_exp = per_range * threads * (long)(sizeof kExp / sizeof kExp[0]);
  printf("{\"fma\": %d, \"via_sincos\": %d, \"sin_args\": %ld, \"cos_args\": %ld, \"exp_args\": %ld, \"sin_bad\": %ld, \"cos_bad\": %ld, \"exp_bad\": %ld}\n",
         fma_variant, g_via_sincos, n_trig, n_trig, n_exp, bad[0].load(), bad[1].load(), bad[2].load());
  static const char *names[] = {"sin", "cos", "exp"};
  for (int f = 0; f < 3; ++f)
    for (int i = 0; i < 4 && i < bad[f].load(); ++i) {
      const double x = first_bad[4 * f + i];
      const double got = f == 0 ? (fma_variant ? sin_<true>(x) : sin_<false>(x)) : f == 1 ? (fma_variant ? cos_<true>(x) : cos_<false>(x))
                                                                                         : (fma_variant ? exp_<true>(x) : exp_<false>(x));
      const double want = f == 0 ? ref_sin(x) : f == 1 ? ref_cos(x) : p_exp(x);
      fprintf(stderr, "%s(%a): restated %a, libm %a\n", names[f], x, got, want);
    }
  return (bad[0].load() || bad[1].load() || bad[2].load()) ? 1 : 0;
}
