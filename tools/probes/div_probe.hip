// scratch: a / b from a reciprocal refined AHEAD (the part of the IEEE division sequence that depends on the
// denominator alone) against the compiler's own division, bit for bit -- for the dependent chains of
// MeanProbabilityCell updates, where the denominator (n + 1) is known a step early.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/div_probe.hip -o tools/_build/div_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

__device__ __forceinline__ double refined_rcp(double y) {
  const double r0 = __builtin_amdgcn_rcp(y);
  const double f0 = __builtin_fma(-y, r0, 1.0);
  const double r1 = __builtin_fma(r0, f0, r0);
  const double f2 = __builtin_fma(-y, r1, 1.0);
  return __builtin_fma(r1, f2, r1);
}
__device__ __forceinline__ bool safe(double x) {
  const double ax = fabs(x);
  return ax > 0x1p-500 && ax < 0x1p500;
}
__device__ __forceinline__ double div_with(double x, double y, double r) {
  if (!(safe(x) && safe(y))) return x / y;
  const double q = x * r;
  const double e = __builtin_fma(-y, q, x);
  const double q2 = __builtin_fma(e, r, q);
  return __builtin_amdgcn_div_fixup(q2, y, x);
}
// ... and without v_div_fixup, which inside the safe range has nothing to fix (no zero, infinity, NaN, no scaled
// operand): one dependent operation less in the chain
__device__ __forceinline__ double div_nofix(double x, double y, double r) {
  const double q = x * r;
  const double e = __builtin_fma(-y, q, x);
  return __builtin_fma(e, r, q);
}
__global__ void k(const double *x, const double *y, int n, unsigned long long *bad, double *ex) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = x[i] / y[i];
  const double b = div_with(x[i], y[i], refined_rcp(y[i]));
  if (__double_as_longlong(a) != __double_as_longlong(b)) {
    if (atomicAdd(bad, 1ull) == 0) { ex[0] = x[i]; ex[1] = y[i]; ex[2] = a; ex[3] = b; }
  }
  if (safe(x[i]) && safe(y[i])) {
    atomicAdd(bad + 2, 1ull);
    const double c = div_nofix(x[i], y[i], refined_rcp(y[i]));
    if (__double_as_longlong(a) != __double_as_longlong(c)) {
      if (atomicAdd(bad + 1, 1ull) == 0) { ex[4] = x[i]; ex[5] = y[i]; ex[6] = a; ex[7] = c; }
    }
  }
}
int main() {
  const int n = 1 << 24;
  std::vector<double> x(n), y(n);
  std::mt19937_64 g(7);
  for (int i = 0; i < n; ++i) {
    const int kind = i & 3;
    if (kind == 0) {  // MEAN-like: (c n + p) / (n + 1)
      const double nn = (double)(g() % 100000), c = (g() >> 11) * 0x1p-53, p = (g() >> 11) * 0x1p-53;
      x[i] = c * nn + p; y[i] = nn + 1;
    } else if (kind == 1) {  // random mantissas, moderate exponents
      uint64_t a = (g() & 0x000fffffffffffffull) | ((uint64_t)(1023 - 40 + g() % 80) << 52), b = (g() & 0x000fffffffffffffull) | ((uint64_t)(1023 - 40 + g() % 80) << 52);
      std::memcpy(&x[i], &a, 8); std::memcpy(&y[i], &b, 8);
    } else if (kind == 2) {  // integers over integers
      x[i] = (double)(g() % 1000000); y[i] = (double)(1 + g() % 1000000);
    } else {  // wide exponents (inside and outside the safe range), signs
      uint64_t a = g(), b = g();
      std::memcpy(&x[i], &a, 8); std::memcpy(&y[i], &b, 8);
    }
  }
  double *dx, *dy, *dex; unsigned long long *dbad;
  hipMalloc(&dx, 8 * n); hipMalloc(&dy, 8 * n); hipMalloc(&dex, 64); hipMalloc(&dbad, 24);
  hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), 8 * n, hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 24);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n, dbad, dex);
  unsigned long long bad3[3] = {0, 0, 0}; double ex[8];
  hipMemcpy(bad3, dbad, 24, hipMemcpyDeviceToHost); hipMemcpy(ex, dex, 64, hipMemcpyDeviceToHost);
  const unsigned long long bad = bad3[0];
  std::printf("%d divisions, %llu different", n, bad);
  if (bad) std::printf(" (first: %a / %a = %a, got %a)", ex[0], ex[1], ex[2], ex[3]);
  std::printf("; without v_div_fixup inside the safe range: %llu of %llu different", bad3[1], bad3[2]);
  if (bad3[1]) std::printf(" (first: %a / %a = %a, got %a)", ex[4], ex[5], ex[6], ex[7]);
  std::printf("\n");
  return (bad || bad3[1]) ? 1 : 0;
}
