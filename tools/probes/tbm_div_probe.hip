// probe (r06): the TBM update's seven quotients from two reciprocals refined ahead, WITHOUT v_div_scale / v_div_fmas /
// v_div_fixup -- q = x r; e = fma(-y, q, x); q' = fma(e, r, q) -- against the compiler's own x / y, bit for bit, over the
// operand ranges mu_wave_apply's TBM fast round establishes: numerators 0 or in [2^-600, 2], denominators in [0.2, 2.5].
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/tbm_div_probe.hip -o /tmp/tbm_div_probe && /tmp/tbm_div_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <random>
#include <vector>

__device__ __forceinline__ double refined_rcp(double y) {
  const double r0 = __builtin_amdgcn_rcp(y);
  const double f0 = __builtin_fma(-y, r0, 1.0);
  const double r1 = __builtin_fma(r0, f0, r0);
  const double f2 = __builtin_fma(-y, r1, 1.0);
  return __builtin_fma(r1, f2, r1);
}
__device__ __forceinline__ double div_nofix(double x, double y, double r) {
  const double q = x * r;
  const double e = __builtin_fma(-y, q, x);
  return __builtin_fma(e, r, q);
}
__global__ void k(const double *x, const double *y, int n, unsigned long long *bad, double *ex) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = x[i] / y[i];
  const double b = div_nofix(x[i], y[i], refined_rcp(y[i]));
  if (__double_as_longlong(a) != __double_as_longlong(b)) {
    if (atomicAdd(bad, 1ull) == 0) { ex[0] = x[i]; ex[1] = y[i]; ex[2] = a; ex[3] = b; }
  }
}
int main() {
  const int n = 1 << 25;
  std::vector<double> x(n), y(n);
  std::mt19937_64 g(11);
  auto u01 = [&]() { return (double)(g() >> 11) * 0x1p-53; };
  for (int i = 0; i < n; ++i) {
    const int kind = i & 7;
    y[i] = 0.2 + 2.3 * u01();
    if (kind == 7) y[i] = 1.0 + (double)((long long)(g() % 2001) - 1000) * 0x1p-52;  // around 1
    if (kind == 0) x[i] = 0.0;
    else if (kind == 1) x[i] = u01();
    else if (kind == 2) x[i] = std::ldexp(0.5 + 0.5 * u01(), -(int)(g() % 600));  // down to 2^-600
    else if (kind == 3) x[i] = y[i] * (1.0 - std::ldexp(u01(), -(int)(g() % 53)));  // quotients just below 1
    else if (kind == 4) x[i] = std::ldexp(u01(), -(int)(g() % 40));
    else if (kind == 5) x[i] = 2.0 * u01();
    else if (kind == 6) x[i] = std::ldexp(1.0, -(int)(g() % 600));  // exact powers of two
    else x[i] = u01();
  }
  double *dx, *dy, *dex;
  unsigned long long *dbad;
  hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dex, 64); hipMalloc(&dbad, 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 8);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n, dbad, dex);
  unsigned long long bad = 0;
  double ex[8];
  hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(ex, dex, 64, hipMemcpyDeviceToHost);
  printf("%d divisions, %llu differ", n, bad);
  if (bad) printf(" (first: %a / %a = %a, got %a)", ex[0], ex[1], ex[2], ex[3]);
  printf("\n");
  return bad ? 1 : 0;
}
