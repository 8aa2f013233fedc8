// sort_probe.hip -- which rocprim configuration sorts K6's (key, beam) pairs fastest on gfx950?
// build here, run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sort_probe.hip -o tools/_build/sort_probe
// Results of r01 (MI355X), 21.6 M random (4-byte key, 4-byte value) pairs, 29 key bits:
//   rocprim default 764 us | onesweep 8 bits, 1024x8, match 627 us (used by the batch) | 512x12 match 692 |
//   256x12 match 952 | 7 bits 512x16 1818 | basic ranking 256x12 3586; 8-byte keys, 36 bits: default 1025 us
//   21 bits only: 492 us; segmented sort over 100 segments of 216 k: 3152 us; 28 bits in three passes of
//   10 bits: 712 us (1024x8), 1067 (512x8), 1406 (256x16) against 625 us for four passes of 8
//   single scan, 22 bits: 170 k pairs merge sort (default) 58 us, onesweep 118 us, merge sort with 4096-item
//   sort blocks 55 us; 600 k: 139 / 135 / 119 us
#include <hip/hip_runtime.h>
#include <string.h>

#include <cstdio>
#include <rocprim/rocprim.hpp>
#include <vector>

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e = (x);                                              \
    if (e != hipSuccess) {                                           \
      printf("%s: %s\n", #x, hipGetErrorString(e));                  \
      return 1;                                                      \
    }                                                                \
  } while (0)

template <typename Config, typename Key>
int run(const char *name, size_t n, unsigned bits, Key *k_in, Key *k_out, unsigned *v_in, unsigned *v_out) {
  size_t tb = 0;
  CK((rocprim::radix_sort_pairs<Config>(nullptr, tb, k_in, k_out, v_in, v_out, n, 0, bits, 0)));
  void *tmp;
  CK(hipMalloc(&tmp, tb));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) CK((rocprim::radix_sort_pairs<Config>(tmp, tb, k_in, k_out, v_in, v_out, n, 0, bits, 0)));
  hipEventRecord(e0, 0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) CK((rocprim::radix_sort_pairs<Config>(tmp, tb, k_in, k_out, v_in, v_out, n, 0, bits, 0)));
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s key %zuB n %zu bits %u: %8.1f us\n", name, sizeof(Key), n, bits, ms / reps * 1e3);
  hipFree(tmp);
  return 0;
}

template <typename Key>
__global__ void fill(Key *k, unsigned *v, size_t n, unsigned bits) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long x = i * 0x9E3779B97F4A7C15ull;
  x ^= x >> 29;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 32;
  k[i] = (Key)(x & ((1ull << bits) - 1));
  v[i] = (unsigned)i;
}

template <unsigned RB, unsigned BS, unsigned IPT,
          rocprim::block_radix_rank_algorithm ALG = rocprim::block_radix_rank_algorithm::default_algorithm>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>,
                                                                           rocprim::kernel_config<BS, IPT>, RB, ALG>>;
constexpr auto kMatch = rocprim::block_radix_rank_algorithm::match;

int main(int argc, char **argv) {
  const bool big = argc > 1;  // any argument: the batch-size experiments too
  const size_t n = 21600000;
  unsigned *k_in, *k_out, *v_in, *v_out;
  CK(hipMalloc(&k_in, 8 * n));
  CK(hipMalloc(&k_out, 8 * n));
  CK(hipMalloc(&v_in, 4 * n));
  CK(hipMalloc(&v_out, 4 * n));
  if (big) {
    fill<unsigned><<<(n + 255) / 256, 256>>>(k_in, v_in, n, 29);
    hipDeviceSynchronize();
    if (run<rocprim::default_config, unsigned>("default", n, 29, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<8, 512, 12, kMatch>, unsigned>("rb8 512x12 match", n, 29, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<8, 1024, 8, kMatch>, unsigned>("rb8 1024x8 match", n, 29, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<8, 1024, 8, kMatch>, unsigned>("rb8 1024x8 match", n, 21, k_in, k_out, v_in, v_out)) return 1;
    // 28 bits in three passes?
    if (run<Cfg<8, 1024, 8, kMatch>, unsigned>("rb8 1024x8 match", n, 28, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<10, 1024, 8, kMatch>, unsigned>("rb10 1024x8 match", n, 28, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<10, 512, 8, kMatch>, unsigned>("rb10 512x8 match", n, 28, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<10, 256, 16, kMatch>, unsigned>("rb10 256x16 match", n, 28, k_in, k_out, v_in, v_out)) return 1;
  }
  // the single-scan size: merge sort (default below 1 M items) against onesweep and bigger sort blocks
  for (size_t m : {(size_t)170000, (size_t)600000}) {
    fill<unsigned><<<(m + 255) / 256, 256>>>(k_in, v_in, m, 22);
    hipDeviceSynchronize();
    if (run<rocprim::default_config, unsigned>("default (merge)", m, 22, k_in, k_out, v_in, v_out)) return 1;
    using Force = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                             rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>,
                                                                                 rocprim::kernel_config<256, 12>, 8>,
                                             4096>;
    if (run<Force, unsigned>("onesweep rb8 256x12", m, 22, k_in, k_out, v_in, v_out)) return 1;
    using M1 = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512, 512, 8>>;
    if (run<M1, unsigned>("merge 512x8", m, 22, k_in, k_out, v_in, v_out)) return 1;
    using M2 = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<1024, 1024, 4>>;
    if (run<M2, unsigned>("merge 1024x4", m, 22, k_in, k_out, v_in, v_out)) return 1;
    using M3 = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<256, 256, 16>>;
    if (run<M3, unsigned>("merge 256x16", m, 22, k_in, k_out, v_in, v_out)) return 1;
    using M5 = rocprim::radix_sort_config<rocprim::default_config,
                                          rocprim::merge_sort_config<512, 512, 8, 128, 256, 8, 1 << 14>>;
    if (run<M5, unsigned>("merge 512x8 mergepath", m, 22, k_in, k_out, v_in, v_out)) return 1;
  }
  return 0;
}
