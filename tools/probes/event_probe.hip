// tools/event_probe.hip -- which HIP-event bracket reports the KERNEL's duration (what rocprofv3
// --kernel-trace reports) for an isolated launch on an otherwise idle stream?
//   A: hipEventRecord / launch / hipEventRecord          (what slamhip_profile_* did in r01)
//   B: hipExtLaunchKernelGGL(start, stop)                (events attached to the dispatch itself)
//   C: in-kernel wall_clock64 span (100 MHz constant clock), ground truth for the kernel body
// Build: hipcc -O2 --offload-arch=gfx950 -o gpurun_out/event_probe tools/event_probe.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("%s -> %s\n", #x, hipGetErrorString(e));                              \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

__global__ void k_spin(unsigned long long ticks, unsigned long long *span) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
  if (threadIdx.x == 0) {
    atomicMin(&span[0], t0);
    atomicMax(&span[1], wall_clock64());
  }
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  unsigned long long *span;
  CK(hipMalloc(&span, 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 50;
  for (double us : {2.0, 8.0, 20.0, 100.0}) {
    const unsigned long long ticks = (unsigned long long)(us * 100.0);  // 100 MHz
    for (int blocks : {1, 300, 4096}) {
      std::vector<float> a, b, c;
      for (int mode = 0; mode < 2; ++mode) {
        for (int r = 0; r < reps; ++r) {
          unsigned long long init[2] = {~0ull, 0ull};
          CK(hipMemcpy(span, init, 16, hipMemcpyHostToDevice));
          CK(hipStreamSynchronize(st));
          if (mode == 0) {
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st, ticks, span);
            CK(hipEventRecord(e1, st));
          } else {
            hipExtLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st, e0, e1, 0, ticks, span);
          }
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          unsigned long long got[2];
          CK(hipMemcpy(got, span, 16, hipMemcpyDeviceToHost));
          (mode == 0 ? a : b).push_back(ms * 1e3f);
          c.push_back((got[1] - got[0]) / 100.0f);
        }
      }
      auto med = [](std::vector<float> v) {
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
      };
      printf("spin %6.1f us x %5d blocks: A record-pair %7.2f us | B ext-launch %7.2f us | C in-kernel %7.2f us\n",
             us, blocks, med(a), med(b), med(c));
    }
  }
  return 0;
}
