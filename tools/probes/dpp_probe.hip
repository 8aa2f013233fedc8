// which lane does each cross-lane primitive read from?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK>
__device__ int dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, BANK, false); }
__global__ void k(int *out) {
  const int lane = threadIdx.x;
  int v = lane;
  out[0 * 64 + lane] = dpp<0xB1, 0xF>(v, v);           // quad_perm [1,0,3,2]
  out[1 * 64 + lane] = dpp<0x4E, 0xF>(v, v);           // quad_perm [2,3,0,1]
  int x4 = dpp<0x124, 0x5>(v, v);                       // row_ror:4 into banks 0,2
  x4 = dpp<0x12C, 0xA>(x4, v);                          // row_ror:12 into banks 1,3
  out[2 * 64 + lane] = x4;
  int y4 = dpp<0x12C, 0x5>(v, v);
  y4 = dpp<0x124, 0xA>(y4, v);
  out[3 * 64 + lane] = y4;
  out[4 * 64 + lane] = dpp<0x128, 0xF>(v, v);          // row_ror:8
  auto r16 = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  out[5 * 64 + lane] = (int)r16[0];
  out[6 * 64 + lane] = (int)r16[1];
  auto r32 = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  out[7 * 64 + lane] = (int)r32[0];
  out[8 * 64 + lane] = (int)r32[1];
}
int main() {
  int *d; hipMalloc(&d, 9 * 64 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[9 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[9] = {"quad 0xB1", "quad 0x4E", "ror4/b5+ror12/bA", "ror12/b5+ror4/bA", "ror8", "pl16swap[0]", "pl16swap[1]", "pl32swap[0]", "pl32swap[1]"};
  for (int r = 0; r < 9; ++r) {
    int x = -2;  // is it a pure xor?
    for (int l = 0; l < 64; ++l) { const int d_ = h[r * 64 + l] ^ l; if (l == 0) x = d_; else if (x != d_) x = -1; }
    printf("%-20s xor %d :", names[r], x);
    for (int l = 0; l < 20; ++l) printf(" %d", h[r * 64 + l]);
    printf(" ... %d %d\n", h[r * 64 + 40], h[r * 64 + 63]);
  }
}
