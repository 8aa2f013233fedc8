// libm_exact.h -- glibc 2.35's sin, cos and exp (x86-64) restated operation for operation, so that the device returns
// the bits the reference's host libm returns.
//
// Why: the reference's DEFAULT trig provider evaluates std::sin / std::cos(theta + a) per beam
// (RawTrigonometryProvider, src/core/trigonometry_utils.h:17-35; use_trig_cache = false, src/ros/init_utils.h:56-58)
// and GmappingBaseCell::discrepancy is 1 - std::exp(-d^2 / 0.05) (src/slams/gmapping/gmapping_grid_cell.h:35-38).
// The device's own sincos / exp differ from glibc's in the last place now and then, which is inside the 1e-5 contract
// but leaves no bit-exact mode for those two configurations.
//
// What is restated (third-party arithmetic absent from /root/reference, SURVEY 8c: glibc 2.35, the toolchain's libm):
//   * sin / cos: the IBM Accurate Mathematical Library as shipped since glibc 2.28 (sysdeps/ieee754/dbl-64/s_sin.c:
//     __sin, __cos, do_sin, do_cos, do_sincos, reduce_sincos, TAYLOR_SIN; constants of usncs.h; table sincostab.c):
//     |x| < 2^-26: x (cos: |x| < 2^-27: 1); |x| < 0.855469: Taylor below 0.126, else a 1/128-spaced table + short
//     polynomials; |x| < 2.426265: pi/2 - |x| through the other function; |x| < 105414350: Cody-Waite reduction by pi/2
//     in four parts.  NOT restated: |x| >= 105414350 (__branred) -- pose heading + beam angle never gets there; those
//     arguments take the platform's sin / cos and are not claimed bit-exact.
//   * exp: Szabolcs Nagy's exp (sysdeps/ieee754/dbl-64/e_exp.c, N = 128 table of e_exp_data.c), every branch.
//
// x86-64 glibc picks an implementation per CPU at load time (ifunc, sysdeps/x86_64/fpu/multiarch): with AVX2 + FMA
// usable the same C source compiled with -mfma -mavx2, i.e. with the compiler's contractions of a * b + c into fused
// multiply-adds; otherwise (no FMA4 machine is considered) the plain build.  The contractions below are the ones in
// Ubuntu's libm-2.35 (0ubuntu3.x), read off its __sin_fma / __cos_fma / __exp_fma; they are uniform: every inlined
// copy of do_sin / do_cos / reduce_sincos / TAYLOR_SIN contracts the same way.  FMA = false evaluates the same
// expression trees with every product rounded (the generic / sse2 build).  Everything here must be compiled with
// -ffp-contract=off (the project's flags): a fused operation appears exactly where it is written.
//
// Pinned by tests/native/libm_exact_check.cpp (CPU suite): both variants against the running libm over > 10^8 arguments
// per function (the FMA-less one with GLIBC_TUNABLES masking FMA / AVX2 out of libm's choice), and on the GPU by
// tests/test_gpu_libm_exact.py: the device's results against the host's, bit for bit.
#pragma once
#include <math.h>
#include <stdint.h>

#include "libm_exact_tables.h"

#if defined(__HIPCC__)
#define SLAMHIP_LIBM_HD __host__ __device__ __forceinline__
#else
#define SLAMHIP_LIBM_HD inline
#endif

namespace slamhip {
namespace libm_exact {

#if defined(__HIPCC__)
__device__ const double d_sincos_tab[440] = {SLAMHIP_LIBM_SINCOS_TABLE};
__device__ const unsigned long long d_exp_tab[256] = {SLAMHIP_LIBM_EXP_TABLE};
#endif
static const double h_sincos_tab[440] = {SLAMHIP_LIBM_SINCOS_TABLE};
static const unsigned long long h_exp_tab[256] = {SLAMHIP_LIBM_EXP_TABLE};

SLAMHIP_LIBM_HD const double *sincos_tab() {
#if defined(__HIP_DEVICE_COMPILE__)
  return d_sincos_tab;
#else
  return h_sincos_tab;
#endif
}
SLAMHIP_LIBM_HD const unsigned long long *exp_tab() {
#if defined(__HIP_DEVICE_COMPILE__)
  return d_exp_tab;
#else
  return h_exp_tab;
#endif
}

SLAMHIP_LIBM_HD uint64_t bits_of(double x) { return (uint64_t)__builtin_bit_cast(unsigned long long, x); }
SLAMHIP_LIBM_HD double double_of(uint64_t b) { return __builtin_bit_cast(double, (unsigned long long)b); }

// a * b + c, c - a * b, a * b - c: fused where glibc's FMA build fuses, else with the product rounded
template <bool FMA>
SLAMHIP_LIBM_HD double mad(double a, double b, double c) {
  if (FMA) return __builtin_fma(a, b, c);
  const double p = a * b;
  return p + c;
}
template <bool FMA>
SLAMHIP_LIBM_HD double nmad(double a, double b, double c) {
  if (FMA) return __builtin_fma(-a, b, c);
  const double p = a * b;
  return c - p;
}
template <bool FMA>
SLAMHIP_LIBM_HD double msub(double a, double b, double c) {
  if (FMA) return __builtin_fma(a, b, -c);
  const double p = a * b;
  return p - c;
}

// ---- sin / cos (s_sin.c) ---------------------------------------------------------------------------------------------
constexpr double kSn3 = -0x1.5555555555515p-3, kSn5 = 0x1.11110e829872fp-7;                         // s_sin.c:57-61
constexpr double kCs2 = 0x1.0000000000000p-1, kCs4 = -0x1.5555555555535p-5, kCs6 = 0x1.6c16bedd9e239p-10;
constexpr double kS1 = -0x1.5555555555555p-3, kS2 = 0x1.1111111110ecep-7, kS3 = -0x1.a01a019db08b8p-13,  // usncs.h
                 kS4 = 0x1.71de27b9a7ed9p-19, kS5 = -0x1.addffc2fcdf59p-26;
constexpr double kBig = 0x1.8p45, kToInt = 0x1.8p52, kHpInv = 0x1.45f306dc9c883p-1;
constexpr double kHp0 = 0x1.921fb54442d18p+0, kHp1 = 0x1.1a62633145c07p-54;
constexpr double kMp1 = 0x1.921fb58000000p+0, kMp2 = -0x1.dde973c000000p-27, kPp3 = -0x1.cb3b398000000p-55,
                 kPp4 = -0x1.d747f23e32ed7p-83;

// TAYLOR_SIN (s_sin.c:71-76): x + ((POLYNOMIAL(xx) x - 0.5 dx) xx + dx), POLYNOMIAL = ((((s5 xx + s4) xx + s3) xx + s2) xx) + s1
template <bool FMA>
SLAMHIP_LIBM_HD double taylor_sin(double x, double dx) {
  const double xx = x * x;
  double p = mad<FMA>(xx, kS5, kS4);
  p = mad<FMA>(xx, p, kS3);
  p = mad<FMA>(xx, p, kS2);
  p = mad<FMA>(xx, p, kS1);
  const double h = 0.5 * dx;
  const double t = mad<FMA>(xx, msub<FMA>(p, x, h), dx);
  return x + t;
}

// do_sin (s_sin.c:125-147): sin(x + dx), |x| < 0.855469
template <bool FMA>
SLAMHIP_LIBM_HD double do_sin(double x, double dx) {
  const double ax = __builtin_fabs(x);
  if (ax < 0.126) return taylor_sin<FMA>(x, dx);
  if (x <= 0) dx = -dx;
  const double u = kBig + ax;
  const double xr = ax - (u - kBig);
  const double *t = sincos_tab() + 4 * (int)(uint32_t)bits_of(u);
  const double xx = xr * xr;
  const double s = xr + mad<FMA>(xr * xx, mad<FMA>(xx, kSn5, kSn3), dx);
  const double q = mad<FMA>(xx, mad<FMA>(xx, kCs6, kCs4), kCs2);
  const double c = mad<FMA>(xr, dx, xx * q);
  const double sn = t[0], ssn = t[1], cs = t[2], ccs = t[3];
  const double cor = mad<FMA>(s, cs, nmad<FMA>(c, sn, mad<FMA>(s, ccs, ssn)));
  return __builtin_copysign(sn + cor, x);
}

// do_cos (s_sin.c:101-119): cos(x + dx), |x| < 0.855469
template <bool FMA>
SLAMHIP_LIBM_HD double do_cos(double x, double dx) {
  if (x < 0) dx = -dx;
  const double ax = __builtin_fabs(x);
  const double u = kBig + ax;
  const double xr = (ax - (u - kBig)) + dx;
  const double *t = sincos_tab() + 4 * (int)(uint32_t)bits_of(u);
  const double xx = xr * xr;
  const double s = mad<FMA>(xr * xx, mad<FMA>(xx, kSn5, kSn3), xr);
  const double c = xx * mad<FMA>(xx, mad<FMA>(xx, kCs6, kCs4), kCs2);
  const double sn = t[0], ssn = t[1], cs = t[2], ccs = t[3];
  const double cor = nmad<FMA>(s, sn, nmad<FMA>(c, cs, nmad<FMA>(s, ssn, ccs)));
  return cs + cor;
}

// reduce_sincos (s_sin.c:153-178): x = n pi/2 + (a + da), |x| < 105414350
template <bool FMA>
SLAMHIP_LIBM_HD int reduce_sincos(double x, double &a, double &da) {
  const double t = mad<FMA>(x, kHpInv, kToInt);
  const double xn = t - kToInt;
  const double y = nmad<FMA>(xn, kMp2, nmad<FMA>(xn, kMp1, x));
  const int n = (int)((uint32_t)bits_of(t) & 3u);
  const double t2 = nmad<FMA>(xn, kPp3, y);
  double db = nmad<FMA>(xn, kPp3, y - t2);
  const double b = nmad<FMA>(xn, kPp4, t2);
  db = db + nmad<FMA>(xn, kPp4, t2 - b);
  a = b;
  da = db;
  return n;
}

// do_sincos (s_sin.c:181-193)
template <bool FMA>
SLAMHIP_LIBM_HD double do_sincos(double a, double da, int n) {
  const double r = (n & 1) ? do_cos<FMA>(a, da) : do_sin<FMA>(a, da);
  return (n & 2) ? -r : r;
}

// the arguments __branred takes (|x| >= 105414350, inf, nan): the platform's function, not claimed bit-exact
SLAMHIP_LIBM_HD double huge_sin(double x) { return ::sin(x); }
SLAMHIP_LIBM_HD double huge_cos(double x) { return ::cos(x); }

// __sin (s_sin.c:200-262)
template <bool FMA>
SLAMHIP_LIBM_HD double sin_(double x) {
  const int k = (int)(uint32_t)(bits_of(x) >> 32) & 0x7fffffff;
  if (k < 0x3e500000) return x;
  if (k < 0x3feb6000) return do_sin<FMA>(x, 0.0);
  if (k < 0x400368fd) {
    const double t = kHp0 - __builtin_fabs(x);
    return __builtin_copysign(do_cos<FMA>(t, kHp1), x);
  }
  if (k < 0x419921FB) {
    double a, da;
    const int n = reduce_sincos<FMA>(x, a, da);
    return do_sincos<FMA>(a, da, n);
  }
  return huge_sin(x);
}

// __cos (s_sin.c:270-330)
template <bool FMA>
SLAMHIP_LIBM_HD double cos_(double x) {
  const int k = (int)(uint32_t)(bits_of(x) >> 32) & 0x7fffffff;
  if (k < 0x3e400000) return 1.0;
  if (k < 0x3feb6000) return do_cos<FMA>(x, 0.0);
  if (k < 0x400368fd) {
    const double y = kHp0 - __builtin_fabs(x);
    const double a = y + kHp1;
    const double da = (y - a) + kHp1;
    return do_sin<FMA>(a, da);
  }
  if (k < 0x419921FB) {
    double a, da;
    const int n = reduce_sincos<FMA>(x, a, da);
    return do_sincos<FMA>(a, da, n + 1);
  }
  return huge_cos(x);
}

// ---- exp (e_exp.c) ---------------------------------------------------------------------------------------------------
constexpr double kInvLn2N = 0x1.71547652b82fep+7, kShift = 0x1.8p52, kNegLn2hiN = -0x1.62e42fefa0000p-8,
                 kNegLn2loN = -0x1.cf79abc9e3b3ap-47;
constexpr double kC2 = 0x1.ffffffffffdbdp-2, kC3 = 0x1.555555555543cp-3, kC4 = 0x1.55555cf172b91p-5, kC5 = 0x1.1111167a4d017p-7;

// specialcase (e_exp.c:42-83): 2^(k/N) (1 + tmp) where the scale alone would over- or underflow (512 <= |x| < 1024)
template <bool FMA>
SLAMHIP_LIBM_HD double exp_specialcase(double tmp, uint64_t sbits, uint64_t ki) {
  if ((ki & 0x80000000u) == 0) {  // k > 0: the exponent of the scale may have overflowed by <= 460
    sbits -= 1009ull << 52;
    const double scale = double_of(sbits);
    return 0x1p1009 * mad<FMA>(scale, tmp, scale);
  }
  sbits += 1022ull << 52;  // k < 0: care for the subnormal range
  const double scale = double_of(sbits);
  const double st = scale * tmp;
  double y = scale + st;
  if (y < 1.0) {  // round to the right bit: the double rounding of a subnormal result
    double lo = (scale - y) + st;
    const double hi = 1.0 + y;
    lo = ((1.0 - hi) + y) + lo;
    y = (hi + lo) - 1.0;
    if (y == 0.0) y = 0.0;  // no -0 under round-to-nearest
  }
  return 0x1p-1022 * y;
}

// __exp (e_exp.c:87-158)
template <bool FMA>
SLAMHIP_LIBM_HD double exp_(double x) {
  const uint64_t ix = bits_of(x);
  uint32_t abstop = (uint32_t)(ix >> 52) & 0x7ffu;
  if (abstop - 0x3c9u >= 0x3fu) {  // |x| < 2^-54, |x| >= 512, inf, nan
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;  // tiny: also +-0
    if (abstop >= 0x409u) {                             // |x| >= 1024
      if (ix == 0xfff0000000000000ull) return 0.0;
      if (abstop >= 0x7ffu) return 1.0 + x;              // +inf, nan
      return (ix >> 63) ? 0x1p-767 * 0x1p-767 : 0x1p769 * 0x1p769;  // __math_uflow / __math_oflow
    }
    abstop = 0;  // 512 <= |x| < 1024: through specialcase
  }
  double kd = mad<FMA>(x, kInvLn2N, kShift);  // z = InvLn2N x; kd = z + Shift
  const uint64_t ki = bits_of(kd);
  kd = kd - kShift;
  const double r = mad<FMA>(kd, kNegLn2loN, mad<FMA>(kd, kNegLn2hiN, x));
  const unsigned long long *T = exp_tab() + 2 * (ki & 127u);
  const double tail = double_of(T[0]);
  const uint64_t sbits = T[1] + (ki << 45);
  const double r2 = r * r;
  // tmp = tail + r + r2 (C2 + r C3) + r2 r2 (C4 + r C5)
  const double tmp = mad<FMA>(r2 * r2, mad<FMA>(r, kC5, kC4), mad<FMA>(mad<FMA>(r, kC3, kC2), r2, tail + r));
  if (abstop == 0) return exp_specialcase<FMA>(tmp, sbits, ki);
  const double scale = double_of(sbits);
  return mad<FMA>(scale, tmp, scale);
}

}  // namespace libm_exact
}  // namespace slamhip
