// Dependent-chain latency of FP64 operations on one wave of gfx950 (the serial cell chains of the map update:
// MeanProbabilityCell 5 dependent operations per observation, TbmBaseCell two normalisations): ns per operation from
// wall_clock64 (100 MHz) over chains of 4096 dependent operations.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fp64_latency_probe.hip -o fp64_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(double *out, long long *ticks, double a, double b) {
  double x = a + threadIdx.x * 1e-9;
  const long long t0 = wall_clock64();
#pragma unroll 1
  for (int i = 0; i < 1024; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KIND == 0) x = x * b;
      if (KIND == 1) x = x + b;
      if (KIND == 2) x = __builtin_fma(x, b, a);
      if (KIND == 3) x = __builtin_amdgcn_rcp(x) + b;      // rcp + add
      if (KIND == 4) x = b / x + a;                       // the compiler's division + add
      if (KIND == 5) x = __builtin_amdgcn_div_fixup(x, b, a) + b;
      if (KIND == 6) x = __builtin_amdgcn_rsq(x) + b;
    }
  }
  const long long t1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) ticks[KIND] = t1 - t0;
}
int main() {
  double *out; long long *ticks;
  hipMalloc(&out, 64 * 8); hipMalloc(&ticks, 8 * 8);
  const char *names[] = {"v_mul_f64", "v_add_f64", "v_fma_f64", "v_rcp_f64 + v_add_f64", "x / y + add", "v_div_fixup_f64 + add", "v_rsq_f64 + add"};
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, ticks, 1.0000001, 0.9999999);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, ticks, 1.0, 1e-9);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, ticks, 1e-9, 0.9999999);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, ticks, 1.5, 0.4);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, ticks, 0.5, 1.25);
    hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, out, ticks, 1.5, 1.25);
    hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, out, ticks, 1.5, 0.4);
    hipDeviceSynchronize();
  }
  long long h[8];
  hipMemcpy(h, ticks, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 7; ++i) std::printf("%-28s %6.1f ns per dependent step\n", names[i], h[i] * 10.0 / 4096.0);
  return 0;
}
