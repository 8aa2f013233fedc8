// tools/latency_probe.hip -- measures the host<->GPU round-trip floor of the matcher's batch loop
// (launch + completion detection variants).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/latency_probe.hip -o /tmp/latency_probe && /tmp/latency_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_empty() {}
__global__ void k_flag(unsigned *counter, unsigned *flag, unsigned seq) {
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// reads one pose per block from `in`, writes one double per block to `out`, then signals
__global__ void k_io(const double *in, double *out, unsigned *counter, unsigned *flag, unsigned seq, int spin) {
  __shared__ double s[3];
  if (threadIdx.x < 3) s[threadIdx.x] = in[3 * blockIdx.x + threadIdx.x];
  __syncthreads();
  double v = s[0] + s[1] + s[2];
  for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9;
  if (threadIdx.x == 0) out[blockIdx.x] = v;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// no per-block fence: results go to coherent host memory with plain stores; completion comes from
// the kernel boundary (stream sync, or a 1-thread follow-up kernel that publishes the flag)
__global__ void k_io_plain(const double *in, double *out, int spin) {
  __shared__ double s[3];
  if (threadIdx.x < 3) s[threadIdx.x] = in[3 * blockIdx.x + threadIdx.x];
  __syncthreads();
  double v = s[0] + s[1] + s[2];
  for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9;
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
__global__ void k_publish(unsigned *flag, unsigned seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// in-kernel completion with RELAXED agent-scope arrival counting: every workgroup drains its own
// stores (s_waitcnt vmcnt(0)), the last one to arrive publishes the flag
__global__ void k_io_count(const double *in, double *out, unsigned *counter, unsigned *flag, unsigned seq, int spin) {
  __shared__ double s[3];
  if (threadIdx.x < 3) s[threadIdx.x] = in[3 * blockIdx.x + threadIdx.x];
  __syncthreads();
  double v = s[0] + s[1] + s[2];
  for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9;
  if (threadIdx.x == 0) {
    out[blockIdx.x] = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned *counter, *flag; CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
  CK(hipHostMalloc(&flag, 4, hipHostMallocMapped | hipHostMallocCoherent)); *flag = 0;
  const int P = 768, IT = 2000;
  double *h_in, *h_out, *d_in, *d_out;
  CK(hipHostMalloc(&h_in, 8 * 3 * P, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostMalloc(&h_out, 8 * P, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipMalloc(&d_in, 8 * 3 * P)); CK(hipMalloc(&d_out, 8 * P));
  for (int i = 0; i < 3 * P; ++i) h_in[i] = i;
  volatile unsigned *vf = flag;
  unsigned seq = 0;
  auto run = [&](const char *name, auto fn) {
    for (int i = 0; i < 50; ++i) fn();
    double t0 = now();
    for (int i = 0; i < IT; ++i) fn();
    printf("%-58s %8.2f us/iter\n", name, (now() - t0) / IT);
  };
  run("empty kernel (768 blk) + hipStreamSynchronize", [&] { hipLaunchKernelGGL(k_empty, dim3(P), dim3(256), 0, st); CK(hipStreamSynchronize(st)); });
  run("flag kernel (768 blk) + host spin on pinned flag", [&] { ++seq; hipLaunchKernelGGL(k_flag, dim3(P), dim3(256), 0, st, counter, flag, seq); while (*vf != seq) __builtin_ia32_pause(); });
  run("flag kernel (6 blk) + host spin", [&] { ++seq; hipLaunchKernelGGL(k_flag, dim3(6), dim3(256), 0, st, counter, flag, seq); while (*vf != seq) __builtin_ia32_pause(); });
  run("io kernel host-in/host-out (768 blk) + spin", [&] { ++seq; hipLaunchKernelGGL(k_io, dim3(P), dim3(256), 0, st, h_in, h_out, counter, flag, seq, 0); while (*vf != seq) __builtin_ia32_pause(); });
  run("io kernel dev-in/host-out (768 blk) + spin (no H2D)", [&] { ++seq; hipLaunchKernelGGL(k_io, dim3(P), dim3(256), 0, st, d_in, h_out, counter, flag, seq, 0); while (*vf != seq) __builtin_ia32_pause(); });
  run("H2D memcpyAsync + io dev-in/host-out + spin", [&] { ++seq; CK(hipMemcpyAsync(d_in, h_in, 8 * 3 * P, hipMemcpyHostToDevice, st)); hipLaunchKernelGGL(k_io, dim3(P), dim3(256), 0, st, d_in, h_out, counter, flag, seq, 0); while (*vf != seq) __builtin_ia32_pause(); });
  run("H2D + io dev-in/dev-out + D2H + hipStreamSynchronize", [&] { ++seq; CK(hipMemcpyAsync(d_in, h_in, 8 * 3 * P, hipMemcpyHostToDevice, st)); hipLaunchKernelGGL(k_io, dim3(P), dim3(256), 0, st, d_in, d_out, counter, (unsigned *)nullptr ? flag : flag, seq, 0); CK(hipMemcpyAsync(h_out, d_out, 8 * P, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
  run("io host-in/host-out with ~7us of work + spin", [&] { ++seq; hipLaunchKernelGGL(k_io, dim3(P), dim3(256), 0, st, h_in, h_out, counter, flag, seq, 2000); while (*vf != seq) __builtin_ia32_pause(); });
  run("io dev-in/host-out with ~7us of work + spin", [&] { ++seq; hipLaunchKernelGGL(k_io, dim3(P), dim3(256), 0, st, d_in, h_out, counter, flag, seq, 2000); while (*vf != seq) __builtin_ia32_pause(); });
  run("plain io host-in/host-out + hipStreamSynchronize", [&] { hipLaunchKernelGGL(k_io_plain, dim3(P), dim3(256), 0, st, h_in, h_out, 0); CK(hipStreamSynchronize(st)); });
  run("plain io host-in/host-out + publish kernel + spin", [&] { ++seq; hipLaunchKernelGGL(k_io_plain, dim3(P), dim3(256), 0, st, h_in, h_out, 0); hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, st, flag, seq); while (*vf != seq) __builtin_ia32_pause(); });
  run("plain io ~7us work + hipStreamSynchronize", [&] { hipLaunchKernelGGL(k_io_plain, dim3(P), dim3(256), 0, st, h_in, h_out, 2000); CK(hipStreamSynchronize(st)); });
  run("plain io ~7us work + publish kernel + spin", [&] { ++seq; hipLaunchKernelGGL(k_io_plain, dim3(P), dim3(256), 0, st, h_in, h_out, 2000); hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, st, flag, seq); while (*vf != seq) __builtin_ia32_pause(); });
  run("plain io dev-in/dev-out ~7us + D2H memcpy + sync", [&] { hipLaunchKernelGGL(k_io_plain, dim3(P), dim3(256), 0, st, d_in, d_out, 2000); CK(hipMemcpyAsync(h_out, d_out, 8 * P, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
  {
    // correctness of the publish-kernel hand-off: every word must be fresh when the flag flips
    int bad = 0;
    for (int it = 0; it < 2000; ++it) {
      ++seq;
      for (int i = 0; i < 3 * P; ++i) h_in[i] = it + i;
      hipLaunchKernelGGL(k_io_plain, dim3(P), dim3(256), 0, st, h_in, h_out, 0);
      hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, st, flag, seq);
      while (*vf != seq) __builtin_ia32_pause();
      for (int b = 0; b < P; ++b) if (h_out[b] != 3.0 * it + 9.0 * b + 3.0) ++bad;
    }
    printf("publish-kernel hand-off: %d stale words in 2000 x %d\n", bad, P);
  }
  run("count io host-in/host-out relaxed arrivals + spin", [&] { ++seq; hipLaunchKernelGGL(k_io_count, dim3(P), dim3(256), 0, st, h_in, h_out, counter, flag, seq, 0); while (*vf != seq) __builtin_ia32_pause(); });
  {
    int bad = 0;
    for (int it = 0; it < 4000; ++it) {
      ++seq;
      for (int i = 0; i < 3 * P; ++i) h_in[i] = it + i;
      hipLaunchKernelGGL(k_io_count, dim3(P), dim3(256), 0, st, h_in, h_out, counter, flag, seq, (it % 7) * 50);
      while (*vf != seq) __builtin_ia32_pause();
      for (int b = 0; b < P; ++b) if (h_out[b] < 3.0 * it + 9.0 * b + 3.0 - 1e-6) ++bad;
    }
    printf("relaxed-arrival hand-off: %d stale words in 4000 x %d\n", bad, P);
  }
  // host writes straight into device memory through the BAR?
  double *fg = nullptr;
  hipError_t e = hipExtMallocWithFlags((void **)&fg, 8 * 3 * P, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
  hipPointerAttribute_t at;
  if (e == hipSuccess && hipPointerGetAttributes(&at, fg) == hipSuccess)
    printf("  type=%d hostPointer=%p devicePointer=%p isManaged=%d\n", (int)at.type, at.hostPointer, at.devicePointer, at.isManaged);
  return 0;
}
