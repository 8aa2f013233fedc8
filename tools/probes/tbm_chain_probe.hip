// The TbmBaseCell update chain of one cell (map_update_kernels.h: mu_step<3> / mu_wave_apply<3>) in isolation: one
// wave applies 1024 observations in order -- every lane the whole step (as k_mu_apply does), or lane k of a quad sum k
// and quotient k (as k_mu_cells' near waves do since r03) -- ns per observation, and the two results compared.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../include -I../../slam-constructor_amd/csrc tbm_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__global__ void k(const double *r0v, const double *r1v, const double *r2v, int n, double *out, long long *ticks) {
  __shared__ double s[3][1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < n; i += 64) { s[0][i] = r0v[i]; s[1][i] = r1v[i]; s[2][i] = r2v[i]; }
  __syncthreads();
  double c0 = 1.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
  long long t0c = wall_clock64();
  for (int t = 0; t < n; ++t) {  // every lane the whole step
    const double r0 = s[0][t], r1 = s[1][t], r2 = s[2][t], r3 = 0.0;
    const double l0 = c0, l1 = c1, l2 = c2, l3 = c3;
    const double t0 = 0.0 + l0 * r0;
    const double t1 = ((0.0 + l0 * r1) + l1 * r0) + l1 * r1;
    const double t2 = ((0.0 + l0 * r2) + l2 * r0) + l2 * r2;
    const double t3 = ((((((((0.0 + l0 * r3) + l1 * r2) + l1 * r3) + l2 * r1) + l2 * r3) + l3 * r0) + l3 * r1) + l3 * r2) + l3 * r3;
    const double tot = t0 + t1 + t2 + t3;
    double n0 = 1.0, n1 = 0.0, n2 = 0.0;
    if (tot != 0.0) { n0 = t0 / tot; n1 = t1 / tot; n2 = t2 / tot; }
    const double w = n0 + n1 + n2;
    if (w == 0.0) { c0 = 1.0; c1 = c2 = 0.0; } else { c0 = n0 / w; c1 = n1 / w; c2 = n2 / w; }
    c3 = 0.0;
  }
  long long t1c = wall_clock64();
  const double a0 = c0, a1 = c1, a2 = c2;
  c0 = 1.0; c1 = 0.0; c2 = 0.0;
  const int role = lane & 3;
  long long t2c = wall_clock64();
  for (int t = 0; t < n; ++t) {  // lane k of a quad: sum k, quotient k
    const double r0 = s[0][t], r1 = s[1][t], r2 = s[2][t];
    const double l0 = c0, l1 = c1, l2 = c2;
    const double x1 = role == 3 ? l1 : l0, y1 = role == 0 ? r0 : (role == 1 ? r1 : r2);
    const double x2 = role == 0 ? 0.0 : (role == 1 ? l1 : l2), y2 = role == 0 ? 0.0 : (role == 3 ? r1 : r0);
    const double x3 = role == 1 ? l1 : (role == 2 ? l2 : 0.0), y3 = role == 1 ? r1 : (role == 2 ? r2 : 0.0);
    const double sum = ((0.0 + x1 * y1) + x2 * y2) + x3 * y3;
    const double t0 = rl(sum, 0), t1 = rl(sum, 1), t2 = rl(sum, 2), t3 = rl(sum, 3);
    const double tot = t0 + t1 + t2 + t3;
    double n0 = 1.0, n1 = 0.0, n2 = 0.0;
    if (tot != 0.0) { const double nq = sum / tot; n0 = rl(nq, 0); n1 = rl(nq, 1); n2 = rl(nq, 2); }
    const double w = n0 + n1 + n2;
    if (w == 0.0) { c0 = 1.0; c1 = c2 = 0.0; } else {
      const double mine = role == 0 ? n0 : (role == 1 ? n1 : n2);
      const double cq = mine / w;
      c0 = rl(cq, 0); c1 = rl(cq, 1); c2 = rl(cq, 2);
    }
  }
  long long t3c = wall_clock64();
  if (lane == 0) {
    ticks[0] = t1c - t0c; ticks[1] = t3c - t2c;
    out[0] = a0; out[1] = a1; out[2] = a2; out[3] = c0; out[4] = c1; out[5] = c2;
  }
}
int main() {
  const int n = 1024;
  double h[3][1024];
  for (int i = 0; i < n; ++i) {
    const double p = (i % 7 == 0) ? 0.95 : 0.01, q = (i % 7 == 0) ? 0.04 : 0.003, eq = q * 1.0;
    const double occ = p * eq, emp = (1 - p) * eq;
    h[0][i] = 1.0 - occ - emp; h[1][i] = emp; h[2][i] = occ;
  }
  double *d0, *d1, *d2, *out; long long *ticks;
  hipMalloc(&d0, 8 * n); hipMalloc(&d1, 8 * n); hipMalloc(&d2, 8 * n); hipMalloc(&out, 64); hipMalloc(&ticks, 16);
  hipMemcpy(d0, h[0], 8 * n, hipMemcpyHostToDevice); hipMemcpy(d1, h[1], 8 * n, hipMemcpyHostToDevice); hipMemcpy(d2, h[2], 8 * n, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d0, d1, d2, n, out, ticks);
  hipDeviceSynchronize();
  long long t[2]; double o[6];
  hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost); hipMemcpy(o, out, 48, hipMemcpyDeviceToHost);
  std::printf("every lane the whole step: %.0f ns per observation; lane-parallel: %.0f ns; results %s (%a %a %a)\n",
              t[0] * 10.0 / n, t[1] * 10.0 / n, (o[0] == o[3] && o[1] == o[4] && o[2] == o[5]) ? "bit-equal" : "DIFFERENT", o[3], o[4], o[5]);
  return 0;
}
