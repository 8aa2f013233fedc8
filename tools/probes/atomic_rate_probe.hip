// Rate of fire-and-forget atomics on random 16-byte-strided words of a large buffer (the try counters of
// k_mu_classify): f64 add, u64 add, u32 add, f32 add, and a plain 8-byte store for comparison.  MI355X.
//   hipcc --offload-arch=gfx950 -O3 atomic_rate_probe.hip -o atomic_rate_probe && ./atomic_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}
template <int KIND>
__global__ __launch_bounds__(1024) void k(double *buf, unsigned long long cells, int per_thread) {
  const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < per_thread; ++i) {
    const unsigned long long at = mix(g * 131 + i) % cells;
    double *p = buf + 2 * at + 1;
    if (KIND == 0) unsafeAtomicAdd(p, 1.0);
    if (KIND == 1) atomicAdd(reinterpret_cast<unsigned long long *>(p), 1ull);
    if (KIND == 2) atomicAdd(reinterpret_cast<unsigned *>(p), 1u);
    if (KIND == 3) unsafeAtomicAdd(reinterpret_cast<float *>(p), 1.0f);
    if (KIND == 4) *p = (double)i;
    if (KIND == 5) atomicAdd(p, 1.0);
  }
}
int main() {
  const unsigned long long cells = 1ull << 28;  // 4 GB of (hits, tries) pairs
  double *buf;
  CK(hipMalloc(&buf, cells * 16));
  CK(hipMemset(buf, 0, cells * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 512 * 8, per = 32;
  const double n = (double)blocks * 1024 * per;
  const char *names[] = {"f64 add (unsafeAtomicAdd)", "u64 add", "u32 add", "f32 add (unsafe)", "plain 8-byte store", "f64 add (atomicAdd)"};
  for (int kind = 0; kind < 6; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      switch (kind) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, buf, cells, per); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, buf, cells, per); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, buf, cells, per); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(1024), 0, 0, buf, cells, per); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(1024), 0, 0, buf, cells, per); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(1024), 0, 0, buf, cells, per); break;
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-28s %8.3f ms for %.0f M operations = %6.1f G/s\n", names[kind], ms, n / 1e6, n / ms / 1e6);
    }
  }
  return 0;
}
