// hc_resident_device.h -- what the co-resident chain kernels (hc_resident.hip: the 1-cell and window OOPEs;
// hc_resident_gm.hip: the GMapping OOPE) share: granule stores / loads, tags, lane broadcasts.
#pragma once

#include <hip/hip_runtime.h>

#include "hc_chain_device.h"

namespace slamhip {
namespace {

constexpr unsigned kHcSpinLimit = 1u << 17;  // most polls of one sweep (~0.4 us each) before a chain gives up; the host
                                             // passes a tighter bound once it has seen matches (HcChainArgs::spin_limit)
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
  m.nbr_ok = p->nbr_ok;
  m.reserved_ = 0;
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

// 16-bit tag of super-step k of a co-resident launch, never 0: twelve bits of step, four of `tag_epoch` = the number of
// co-resident launches on this exchange block so far.  Four are enough because every workgroup clears its own
// granules when a launch starts: what can still lie in a slot is the previous LAUNCH's, whose bits differ by one (the
// match epoch would not do: matches run in other forms in between bump it too, and sixteen of those would bring the
// same bits back -- ADVICE r4; the host clears the block when a launch uses more slots than the one before).  The
// granule's last dword = 16 fingerprint bits | tag.
__device__ __forceinline__ unsigned hc_tag(unsigned tag_epoch, int k) { return ((tag_epoch & 0xfu) << 12) | (unsigned)(k + 1); }

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
// G granules at once: the loads AND their one wait in ONE asm statement with early-clobber outputs, so that the
// compiler can neither copy, spill nor rename a destination register between a load and the wait it cannot see
// (SIInsertWaitcnts does not look inside inline asm: ADVICE r4).  Every address is read: callers clamp the slots
// a sweep does not need to one it does.
__device__ __forceinline__ void gran_fetch(u32x4 (&g)[1], const HcGranule *const (&p)[1]) {
  asm volatile(
               "global_load_dwordx4 %0, %1, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(g[0])
               : "v"(p[0])
               : "memory");
}
__device__ __forceinline__ void gran_fetch(u32x4 (&g)[2], const HcGranule *const (&p)[2]) {
  asm volatile(
               "global_load_dwordx4 %0, %2, off sc1\n\t"
               "global_load_dwordx4 %1, %3, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(g[0]), "=&v"(g[1])
               : "v"(p[0]), "v"(p[1])
               : "memory");
}
__device__ __forceinline__ void gran_fetch(u32x4 (&g)[4], const HcGranule *const (&p)[4]) {
  asm volatile(
               "global_load_dwordx4 %0, %4, off sc1\n\t"
               "global_load_dwordx4 %1, %5, off sc1\n\t"
               "global_load_dwordx4 %2, %6, off sc1\n\t"
               "global_load_dwordx4 %3, %7, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3])
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
               : "memory");
}
__device__ __forceinline__ void gran_fetch(u32x4 (&g)[7], const HcGranule *const (&p)[7]) {
  asm volatile(
               "global_load_dwordx4 %0, %7, off sc1\n\t"
               "global_load_dwordx4 %1, %8, off sc1\n\t"
               "global_load_dwordx4 %2, %9, off sc1\n\t"
               "global_load_dwordx4 %3, %10, off sc1\n\t"
               "global_load_dwordx4 %4, %11, off sc1\n\t"
               "global_load_dwordx4 %5, %12, off sc1\n\t"
               "global_load_dwordx4 %6, %13, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6])
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6])
               : "memory");
}
__device__ __forceinline__ void gran_fetch(u32x4 (&g)[8], const HcGranule *const (&p)[8]) {
  asm volatile(
               "global_load_dwordx4 %0, %8, off sc1\n\t"
               "global_load_dwordx4 %1, %9, off sc1\n\t"
               "global_load_dwordx4 %2, %10, off sc1\n\t"
               "global_load_dwordx4 %3, %11, off sc1\n\t"
               "global_load_dwordx4 %4, %12, off sc1\n\t"
               "global_load_dwordx4 %5, %13, off sc1\n\t"
               "global_load_dwordx4 %6, %14, off sc1\n\t"
               "global_load_dwordx4 %7, %15, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7])
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
               : "memory");
}
__device__ __forceinline__ double gran_score(const u32x4 &g) {
  return __longlong_as_double((long long)(((unsigned long long)g.y << 32) | (unsigned long long)g.x));
}


}  // namespace
}  // namespace slamhip
