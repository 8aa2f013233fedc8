// tools/resident_probe.hip -- round-trip floor of a RESIDENT scoring kernel fed through a pinned
// host mailbox (no launch per batch), against the launch-per-batch loop the matchers used in r01.
//   host: write poses + n, bump cmd_seq  ->  block 0 polls cmd_seq (system scope), republishes it at
//   agent scope  ->  all blocks claim poses from an atomic ticket counter, "score" them (spin ~work
//   us), count completions  ->  the block that completes the last pose copies the scores to host
//   memory, fences, bumps done_seq  ->  host spins on done_seq.
// Every device spin has a wall-clock timeout so a lost host cannot hang the GPU.
//   hipcc --offload-arch=gfx950 -O3 tools/resident_probe.hip -o /tmp/resident_probe && /tmp/resident_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                      \
  do {                                                             \
    hipError_t e = (x);                                            \
    if (e != hipSuccess) {                                         \
      printf("%s: %s\n", #x, hipGetErrorString(e));                \
      exit(1);                                                     \
    }                                                              \
  } while (0)

constexpr unsigned kExit = 0xffffffffu;
constexpr int kMaxPoses = 8192;

struct Mailbox {  // pinned, coherent host memory
  unsigned cmd_seq, n, pad0[14];
  unsigned done_seq, error, pad1[14];
  double poses[3 * kMaxPoses];
  double scores[kMaxPoses];
};
struct DevState {
  unsigned cmd_seq, n, abort, pad;
  unsigned long long next[4], done[4];
  double scores[kMaxPoses];
};

#define LD(p, scope) __hip_atomic_load(p, __ATOMIC_ACQUIRE, scope)
#define SYS __HIP_MEMORY_SCOPE_SYSTEM
#define AGT __HIP_MEMORY_SCOPE_AGENT

__global__ void __launch_bounds__(256) k_resident(Mailbox *hm, DevState *ds, unsigned long long timeout_ticks,
                                                  int work_ticks) {
  __shared__ unsigned s_seq, s_n, s_last;
  __shared__ unsigned long long s_ticket;
  unsigned last_seq = 0;
  const unsigned long long t_begin = wall_clock64();
  for (;;) {
    if (threadIdx.x == 0) {
      unsigned seq = last_seq;
      if (blockIdx.x == 0) {
        while ((seq = LD(&hm->cmd_seq, SYS)) == last_seq) {
          if (wall_clock64() - t_begin > timeout_ticks) {
            __hip_atomic_store(&ds->abort, 1u, __ATOMIC_RELAXED, AGT);
            seq = kExit;
            break;
          }
        }
        if (seq != kExit) {
          const unsigned n = hm->n;
          ds->n = n;
          __hip_atomic_store(&ds->next[seq & 3], (unsigned long long)seq << 32, __ATOMIC_RELAXED, AGT);
          __hip_atomic_store(&ds->done[seq & 3], 0ull, __ATOMIC_RELAXED, AGT);
        }
        __hip_atomic_store(&ds->cmd_seq, seq, __ATOMIC_RELEASE, AGT);
      } else {
        while ((seq = LD(&ds->cmd_seq, AGT)) == last_seq) {
          __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t_begin > timeout_ticks) {
            seq = kExit;
            break;
          }
        }
      }
      s_seq = seq;
      s_n = ds->n;
    }
    __syncthreads();
    const unsigned seq = s_seq, n = s_n;
    if (seq == kExit) return;
    last_seq = seq;
    for (;;) {
      if (threadIdx.x == 0)
        s_ticket = __hip_atomic_fetch_add(&ds->next[seq & 3], 1ull, __ATOMIC_RELAXED, AGT);
      __syncthreads();
      const unsigned long long t = s_ticket;
      const unsigned idx = (unsigned)t;
      if ((unsigned)(t >> 32) != seq) {  // ticket of another command: protocol broken
        if (threadIdx.x == 0) __hip_atomic_store(&hm->error, 1u, __ATOMIC_RELAXED, SYS);
        break;
      }
      if (idx >= n) break;
      // "score": read the pose from the mailbox (PCIe), burn work_ticks, write one double
      double v = 0;
      if (threadIdx.x < 3) v = hm->poses[3 * idx + threadIdx.x];
      v += __shfl_down(v, 1) + __shfl_down(v, 2);
      const unsigned long long w0 = wall_clock64();
      while (wall_clock64() - w0 < (unsigned long long)work_ticks) {
      }
      if (threadIdx.x == 0) {
        ds->scores[idx] = v;
        const unsigned long long prev = __hip_atomic_fetch_add(&ds->done[seq & 3], 1ull, __ATOMIC_ACQ_REL, AGT);
        s_last = (prev == n - 1);
      }
      __syncthreads();
      if (s_last) {
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
          hm->scores[i] = __hip_atomic_load(&ds->scores[i], __ATOMIC_RELAXED, AGT);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&hm->done_seq, seq, __ATOMIC_RELEASE, SYS);
      }
      __syncthreads();
    }
  }
}

// launch-per-batch reference: one block per pose, then a 1-thread publish kernel (r01 matchers)
__global__ void __launch_bounds__(256) k_batch(const double *poses, double *scores, int work_ticks) {
  double v = 0;
  if (threadIdx.x < 3) v = poses[3 * blockIdx.x + threadIdx.x];
  v += __shfl_down(v, 1) + __shfl_down(v, 2);
  const unsigned long long w0 = wall_clock64();
  while (wall_clock64() - w0 < (unsigned long long)work_ticks) {
  }
  if (threadIdx.x == 0) scores[blockIdx.x] = v;
}
__global__ void k_publish(unsigned *flag, unsigned seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, SYS);
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  Mailbox *hm;
  CK(hipHostMalloc((void **)&hm, sizeof(Mailbox), hipHostMallocCoherent | hipHostMallocMapped));
  memset(hm, 0, sizeof(Mailbox));
  DevState *ds;
  CK(hipMalloc(&ds, sizeof(DevState)));
  volatile unsigned *done = &hm->done_seq;
  const int rounds = 300;
  for (int work_us : {0, 5}) {
    for (int n : {6, 200, 1000, 4000}) {
      for (int grid : {256, 512, 1024}) {
        CK(hipMemset(ds, 0, sizeof(DevState)));
        hm->cmd_seq = 0;
        hm->done_seq = 0;
        hm->error = 0;
        __sync_synchronize();
        hipLaunchKernelGGL(k_resident, dim3(grid), dim3(256), 0, st, hm, ds, 100ull * 2000000ull /* 2 s */,
                           work_us * 100);
        std::vector<double> lat;
        bool ok = true;
        for (int r = 1; r <= rounds && ok; ++r) {
          const double t0 = now_us();
          for (int i = 0; i < 3 * n; ++i) hm->poses[i] = r + i;
          hm->n = n;
          __atomic_store_n(&hm->cmd_seq, (unsigned)r, __ATOMIC_RELEASE);
          while (*done != (unsigned)r) {
            if (now_us() - t0 > 1e6) {
              printf("TIMEOUT round %d (n %d grid %d) error %u\n", r, n, grid, hm->error);
              ok = false;
              break;
            }
          }
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
          lat.push_back(now_us() - t0);
          const double expect = 3.0 * r + 3.0 * (n - 1) + 0 + 1 + 2 - 0;  // pose n-1: r+3(n-1)+{0,1,2}
          if (ok && hm->scores[n - 1] != 3.0 * r + 9.0 * (n - 1) + 3.0) {
            printf("BAD score round %d: %f (expect %f)\n", r, hm->scores[n - 1], 3.0 * r + 9.0 * (n - 1) + 3.0);
            (void)expect;
            ok = false;
          }
        }
        __atomic_store_n(&hm->cmd_seq, kExit, __ATOMIC_RELEASE);
        CK(hipStreamSynchronize(st));
        std::sort(lat.begin(), lat.end());
        if (!lat.empty())
          printf("resident  work %d us  n %5d  grid %4d: median %7.2f us  p90 %7.2f  min %7.2f  err %u\n", work_us, n,
                 grid, lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat[0], hm->error);
      }
      // launch-per-batch
      {
        std::vector<double> lat;
        for (int r = 1; r <= rounds; ++r) {
          const double t0 = now_us();
          for (int i = 0; i < 3 * n; ++i) hm->poses[i] = r + i;
          hipLaunchKernelGGL(k_batch, dim3(n), dim3(256), 0, st, hm->poses, hm->scores, work_us * 100);
          hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, st, &hm->done_seq, (unsigned)(r + 100000));
          while (*done != (unsigned)(r + 100000)) {
          }
          lat.push_back(now_us() - t0);
        }
        std::sort(lat.begin(), lat.end());
        printf("per-batch work %d us  n %5d           : median %7.2f us  p90 %7.2f  min %7.2f\n", work_us, n,
               lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat[0]);
      }
      // launch-per-batch, completion published by a stream memory operation instead of a kernel
      {
        std::vector<double> lat;
        bool ok = true;
        for (int r = 1; r <= rounds && ok; ++r) {
          const double t0 = now_us();
          for (int i = 0; i < 3 * n; ++i) hm->poses[i] = r + i;
          hipLaunchKernelGGL(k_batch, dim3(n), dim3(256), 0, st, hm->poses, hm->scores, work_us * 100);
          hipError_t e = hipStreamWriteValue32(st, (void *)&hm->done_seq, (uint32_t)(r + 200000), 0);
          if (e != hipSuccess) {
            printf("hipStreamWriteValue32: %s\n", hipGetErrorString(e));
            ok = false;
            break;
          }
          while (*done != (unsigned)(r + 200000)) {
            if (now_us() - t0 > 1e6) {
              printf("write-value TIMEOUT\n");
              ok = false;
              break;
            }
          }
          lat.push_back(now_us() - t0);
          if (ok && hm->scores[n - 1] != 3.0 * r + 9.0 * (n - 1) + 3.0) {
            printf("write-value STALE score round %d\n", r);
            ok = false;
          }
        }
        std::sort(lat.begin(), lat.end());
        if (ok)
          printf("write-val work %d us  n %5d           : median %7.2f us  p90 %7.2f  min %7.2f\n", work_us, n,
                 lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat[0]);
      }
    }
  }
  return 0;
}
