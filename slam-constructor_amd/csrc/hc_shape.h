// hc_shape.h -- host side of the device-resident hill climbing (hc_chain.h): the static speculation
// shapes.  Header-only so that tests/native/hc_chain_test.cpp builds it without the HIP runtime.
#pragma once

#include <algorithm>
#include <cstring>
#include <vector>

#include "hc_chain.h"

namespace slamhip {

// Static speculation shape for one per-candidate acceptance rate: round instances chosen best-first by
// the probability of reaching them (SpecTree::build_rounds without poses).  repeat_boost: weight of the
// outcome that repeats the parent's move.
inline void hc_build_shape(double p_accept, double repeat_boost, double min_reach, int max_inst, HcShape *out) {
  std::memset(out, 0, sizeof(*out));
  struct Cand {
    double prio;
    int parent, outcome;
    bool operator<(const Cand &o) const { return prio < o.prio; }
  };
  std::vector<Cand> heap;
  heap.push_back(Cand{1.0, -1, -1});
  const double q = 1.0 - p_accept;
  double qk[7], p_out[7];
  qk[0] = 1.0;
  for (int k = 1; k <= 6; ++k) qk[k] = qk[k - 1] * q;
  p_out[0] = qk[6];                                             // all six rejected
  for (int j = 1; j <= 6; ++j) p_out[j] = p_accept * qk[6 - j];  // candidate j-1 accepted last
  if (max_inst > kHcMaxInst) max_inst = kHcMaxInst;
  int n = 0;
  while (!heap.empty() && n < max_inst) {
    if (n > 0 && heap.front().prio < min_reach) break;
    std::pop_heap(heap.begin(), heap.end());
    const Cand cd = heap.back();
    heap.pop_back();
    HcInst in;
    std::memset(&in, 0, sizeof(in));
    for (int o = 0; o < 7; ++o) hc_set_child(in, o, -1);
    hc_set_bp_inst(in, -1);
    hc_set_byte9(in, 7, cd.parent < 0 ? 255u : (unsigned)cd.parent);
    if (cd.parent < 0) {
      hc_set_byte9(in, 5, 1);  // is_root
    } else {
      const HcInst &pr = out->inst[cd.parent];
      if (cd.outcome > 0 && hc_nseg(pr) >= kHcMaxSeg) continue;  // path too long for the record
      if (hc_nfail(pr) >= 250 || hc_depth(pr) >= 250) continue;
      for (int o = 0; o < 7; ++o) in.w[o] = pr.w[o];
      in.w[cd.outcome] |= 1ull << cd.parent;
      in.w[10] = pr.w[10];
      in.w[11] = pr.w[11];
      in.w[12] = pr.w[12];
      in.w[13] = pr.w[13];
      hc_set_byte9(in, 3, hc_depth(pr) + 1);
      hc_set_byte9(in, 2, hc_nfail(pr));
      hc_set_byte9(in, 1, hc_nfail(pr) + (cd.outcome == 0 ? 1 : 0));
      if (cd.outcome == 0) {
        hc_set_byte9(in, 4, hc_nseg(pr));
        hc_set_byte9(in, 6, hc_tail_fail(pr) + 1);
        hc_set_bp_inst(in, hc_bp_inst(pr));
        hc_set_byte9(in, 0, hc_bp_cand(pr));
      } else {
        hc_set_seg(in, hc_nseg(pr), hc_tail_fail(pr), (unsigned)cd.outcome);
        hc_set_byte9(in, 4, hc_nseg(pr) + 1);
        hc_set_byte9(in, 6, 0);
        hc_set_bp_inst(in, cd.parent);
        hc_set_byte9(in, 0, cd.outcome - 1);
      }
      hc_set_child(out->inst[cd.parent], cd.outcome, n);
    }
    out->inst[n] = in;
    const int me = n++;
    double w[7], tot = 0;
    for (int j = 0; j <= 6; ++j) {
      w[j] = p_out[j] * ((j > 0 && j == cd.outcome) ? repeat_boost : 1.0);
      tot += w[j];
    }
    for (int j = 0; j <= 6; ++j) {
      const double prio = cd.prio * w[j] / tot;
      if (prio < min_reach) continue;
      heap.push_back(Cand{prio, me, j});
      std::push_heap(heap.begin(), heap.end());
    }
  }
  out->n_inst = n;
}


}  // namespace slamhip
