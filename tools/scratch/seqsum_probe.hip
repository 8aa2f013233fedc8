// how long does ONE lane need for n dependent f64 adds out of LDS?  (the reference's beam-order sum)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double *in, double *out, long long *cyc, int n) {
  extern __shared__ double s[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = in[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    double acc = 0.0;
    int b = 0;
    for (; b + 8 <= n; b += 8) {
      const double t0_ = s[b], t1 = s[b + 1], t2 = s[b + 2], t3 = s[b + 3], t4 = s[b + 4], t5 = s[b + 5], t6 = s[b + 6], t7 = s[b + 7];
      acc = acc + t0_; acc = acc + t1; acc = acc + t2; acc = acc + t3; acc = acc + t4; acc = acc + t5; acc = acc + t6; acc = acc + t7;
    }
    for (; b < n; ++b) acc = acc + s[b];
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    out[blockIdx.x] = acc;
    if (blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = c1 - c0; }
  }
}
int main() {
  const int n = 1080;
  double *in, *out; long long *cyc;
  hipMalloc(&in, n * 8); hipMalloc(&out, 4096 * 8); hipMalloc(&cyc, 16);
  double h[n]; for (int i = 0; i < n; ++i) h[i] = 1.0 / n * (0.3 + 0.001 * i);
  hipMemcpy(in, h, n * 8, hipMemcpyHostToDevice);
  for (int blocks : {1, 385, 1024}) for (int nt : {64, 256, 1024}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, nt, n * 8>>>(in, out, cyc, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) k<<<blocks, nt, n * 8>>>(in, out, cyc, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("blocks %4d threads %4d: kernel %.2f us, sum loop %lld wall ticks (100 MHz => %.2f us), %lld shader clocks (%.1f per add)\n",
           blocks, nt, ms * 1e3 / 20, c[0], c[0] / 100.0, c[1], (double)c[1] / n);
  }
  return 0;
}
