// hc_resident_device.h -- what the co-resident chain kernels (hc_resident.hip: the 1-cell and window OOPEs;
// hc_resident_gm.hip: the GMapping OOPE) share: granule stores / loads, tags, lane broadcasts.
#pragma once

#include <hip/hip_runtime.h>

#include "hc_chain_device.h"

namespace slamhip {
namespace {

constexpr unsigned kHcSpinLimit = 1u << 17;  // polls of one sweep (~0.4 us each) before the chain gives up
constexpr int kHcResidentMaxSteps = 4000;    // super-steps a tag can count (12 bits, 0 excluded)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int bcast_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ long long bcast_ll(long long v, int lane) {
  const int lo = bcast_i((int)(unsigned)(unsigned long long)v, lane);
  const int hi = bcast_i((int)(unsigned)((unsigned long long)v >> 32), lane);
  return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double bcast(double v, int lane) {
  return __longlong_as_double(bcast_ll(__double_as_longlong(v), lane));
}

// The views read through a constant-address-space pointer (the kernarg segment, or a job table in HBM): scalar
// loads where they are needed instead of scalar registers held -- and spilled -- across a persistent loop
typedef const __attribute__((address_space(4))) MapView *MapViewCP;
typedef const __attribute__((address_space(4))) ScanView *ScanViewCP;
__device__ __forceinline__ MapView load_view(MapViewCP p) {
  MapView m;
  m.payload = p->payload;
  m.width = p->width;
  m.height = p->height;
  m.pitch = p->pitch;
  m.origin_x = p->origin_x;
  m.origin_y = p->origin_y;
  m.scale = p->scale;
  m.inv_scale = p->inv_scale;
  for (int k = 0; k < 4; ++k) m.unknown[k] = p->unknown[k];
  return m;
}
__device__ __forceinline__ ScanView load_view(ScanViewCP p) {
  ScanView s;
  s.range = p->range;
  s.cos_a = p->cos_a;
  s.sin_a = p->sin_a;
  s.weight = p->weight;
  s.factor = p->factor;
  s.n = p->n;
  s.tot_w = p->tot_w;
  return s;
}

// 16-bit tag of super-step k of the match with this epoch, never 0: twelve bits of step, four of epoch.  Four are
// enough because every workgroup clears its own two granules when a match starts: what can still lie in a slot is
// the previous match's, whose epoch bits differ (the host clears the block when a launch uses more slots than
// the one before, hc_resident_capacity's callers).  The granule's last dword = 16 fingerprint bits | tag.
__device__ __forceinline__ unsigned hc_tag(unsigned epoch, int k) { return ((epoch & 0xfu) << 12) | (unsigned)(k + 1); }

// hash: the low 48 bits count (fold_fingerprint48)
__device__ __forceinline__ void gran_store(HcGranule *p, double score, unsigned long long hash, unsigned tag) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(score);
  u32x4 g;
  g.x = (unsigned)u;
  g.y = (unsigned)(u >> 32);
  g.z = (unsigned)hash;
  g.w = ((unsigned)(hash >> 32) << 16) | (tag & 0xffffu);
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ unsigned gran_tag(const u32x4 &g) { return g.w & 0xffffu; }
__device__ __forceinline__ unsigned long long gran_hash(const u32x4 &g) {
  return ((unsigned long long)(g.w >> 16) << 32) | (unsigned long long)g.z;
}
// issue only: gran_wait ties the loaded values to the one wait
__device__ __forceinline__ u32x4 gran_load(const HcGranule *p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// (one operand per granule: the values cannot be used before the wait)
__device__ __forceinline__ void gran_wait(u32x4 (&g)[7]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6])::"memory");
}
__device__ __forceinline__ void gran_wait(u32x4 (&g)[4]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3])::"memory");
}
__device__ __forceinline__ void gran_wait(u32x4 (&g)[1]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0])::"memory");
}
__device__ __forceinline__ void gran_wait(u32x4 (&g)[2]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1])::"memory");
}
__device__ __forceinline__ double gran_score(const u32x4 &g) {
  return __longlong_as_double((long long)(((unsigned long long)g.y << 32) | (unsigned long long)g.x));
}


}  // namespace
}  // namespace slamhip
