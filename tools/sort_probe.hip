// sort_probe.hip -- which rocprim onesweep configuration sorts K6's (key, beam) pairs fastest on gfx950?
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sort_probe.hip -o /tmp/sort_probe && /tmp/sort_probe
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
  printf("%-28s key %zuB n %zu bits %u: %8.1f us  (%.2f TB/s per 2x(key+value) pass-equivalent)\n", name, sizeof(Key), n,
         bits, ms / reps * 1e3, 0.0);
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

int main() {
  const size_t n = 21600000;
  unsigned *k_in, *k_out, *v_in, *v_out;
  CK(hipMalloc(&k_in, 8 * n));
  CK(hipMalloc(&k_out, 8 * n));
  CK(hipMalloc(&v_in, 4 * n));
  CK(hipMalloc(&v_out, 4 * n));
  for (unsigned bits : {29u}) {
    fill<unsigned><<<(n + 255) / 256, 256>>>(k_in, v_in, n, bits);
    hipDeviceSynchronize();
    if (run<rocprim::default_config, unsigned>("default", n, bits, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<8, 256, 12>, unsigned>("rb8 256x12", n, bits, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<8, 256, 16>, unsigned>("rb8 256x16", n, bits, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<8, 512, 12, kMatch>, unsigned>("rb8 512x12 match", n, bits, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<8, 1024, 8, kMatch>, unsigned>("rb8 1024x8 match", n, bits, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<7, 512, 16>, unsigned>("rb7 512x16", n, bits, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<8, 256, 12, kMatch>, unsigned>("rb8 256x12 match", n, bits, k_in, k_out, v_in, v_out)) return 1;
  }
  {
    unsigned long long *k64 = (unsigned long long *)k_in, *k64o = (unsigned long long *)k_out;
    fill<unsigned long long><<<(n + 255) / 256, 256>>>(k64, v_in, n, 36);
    hipDeviceSynchronize();
    if (run<rocprim::default_config, unsigned long long>("default", n, 36, k64, k64o, v_in, v_out)) return 1;
    if (run<Cfg<8, 256, 12>, unsigned long long>("rb8 256x12", n, 36, k64, k64o, v_in, v_out)) return 1;
  }
  {  // the batch as 100 segments (the records arrive grouped by job): segmented sort by the 21 cell bits,
     // against one global stable sort by the cell bits alone (chains then come out as (cell, job))
    const unsigned segs = 100;
    std::vector<unsigned> h_off(segs + 1);
    for (unsigned k = 0; k <= segs; ++k) h_off[k] = (unsigned)((unsigned long long)n * k / segs);
    unsigned *d_off;
    CK(hipMalloc(&d_off, sizeof(unsigned) * (segs + 1)));
    CK(hipMemcpy(d_off, h_off.data(), sizeof(unsigned) * (segs + 1), hipMemcpyHostToDevice));
    fill<unsigned><<<(n + 255) / 256, 256>>>(k_in, v_in, n, 21);
    hipDeviceSynchronize();
    size_t tb = 0;
    CK(rocprim::segmented_radix_sort_pairs(nullptr, tb, k_in, k_out, v_in, v_out, n, segs, d_off, d_off + 1, 0, 21, 0));
    void *tmp;
    CK(hipMalloc(&tmp, tb));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    CK(rocprim::segmented_radix_sort_pairs(tmp, tb, k_in, k_out, v_in, v_out, n, segs, d_off, d_off + 1, 0, 21, 0));
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r)
      CK(rocprim::segmented_radix_sort_pairs(tmp, tb, k_in, k_out, v_in, v_out, n, segs, d_off, d_off + 1, 0, 21, 0));
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("segmented 100 x %zu, 21 bits: %8.1f us\n", n / segs, ms / 5 * 1e3);
    if (run<Cfg<8, 1024, 8, kMatch>, unsigned>("global, cell bits only", n, 21, k_in, k_out, v_in, v_out)) return 1;
    if (run<Cfg<7, 1024, 8, kMatch>, unsigned>("global rb7, cell bits only", n, 21, k_in, k_out, v_in, v_out)) return 1;
  }
  // the single-scan size: merge sort (default below 1M items) against onesweep
  for (size_t m : {(size_t)200000, (size_t)600000}) {
    fill<unsigned><<<(m + 255) / 256, 256>>>(k_in, v_in, m, 22);
    hipDeviceSynchronize();
    if (run<rocprim::default_config, unsigned>("default (merge)", m, 22, k_in, k_out, v_in, v_out)) return 1;
    using Force = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                             rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>,
                                                                                 rocprim::kernel_config<256, 12>, 8>,
                                             4096>;
    if (run<Force, unsigned>("onesweep rb8 256x12", m, 22, k_in, k_out, v_in, v_out)) return 1;
    using Force2 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 4>,
                                                                                  rocprim::kernel_config<256, 4>, 8>,
                                              4096>;
    if (run<Force2, unsigned>("onesweep rb8 256x4", m, 22, k_in, k_out, v_in, v_out)) return 1;
  }
  return 0;
}
