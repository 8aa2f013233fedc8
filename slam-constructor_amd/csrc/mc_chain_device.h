// mc_chain_device.h -- what the Monte-Carlo chain kernel (mc_chain.hip) and its host driver (matchers.cpp) share.
#pragma once

#include "hc_chain_device.h"
#include "mc_chain.h"
#include "slamhip_internal.h"

namespace slamhip {

// Most candidates a super-step speculates on: kMcPerLane per lane of the replaying wave, one short of 64 x 8 so that
// candidates + the bookkeeping workgroup are 512 workgroups of 512 threads -- two on EVERY CU of an MI355X; the launch
// itself keeps one CU's worth of workgroups spare (mc_resident_capacity, ADVICE r4: per_cu x (cus - 1), minus the
// bookkeeping workgroup), so what runs on 256 CUs is 509 candidates + the bookkeeping workgroup (r04 ran
// 384 + 1: half the CUs scored two poses per super-step and set the pace, the other half one; the extra 127
// candidates cost no time per super-step and a Monte-Carlo chain is mostly long runs of rejections).
constexpr int kMcPerLane = 8;
constexpr int kMcSlots = 64 * kMcPerLane - 1;

// one scorer call of the walked path, in the reference's order (what GridScanMatcherObserver sees)
struct McTraceEntry {
  double x, y, theta, score;
  int accepted, pad;
};

// pinned, host-coherent: written by the bookkeeping workgroup, read by the spinning host
struct McHostOut {
  double pose[3];
  double best_prob;
  long long calls, evaluated;
  int steps;
  unsigned rescored;
  // the enumerator's state after the match (GaussianPoseEnumerator: matchers.h)
  long long tape_pos;  // pairs consumed, relative to the uploaded window
  unsigned failed, poses;
  double td, rd;
  int has_saved;
  double saved[3];
  int error;          // 2: trace buffer too small
  unsigned progress;  // super-steps started so far in this process_scan
  unsigned done_seq;  // = epoch of the process_scan whose result is above
};

// device memory of one matcher
struct McChainCtl {
  McState state[2];  // state of super-step k at [k & 1]
  double scores[2][kMcSlots + 8];
  double scores_seq[2][kMcSlots + 8];
  unsigned long long hashes[2][kMcSlots + 8];
  unsigned done_epoch;
};

struct McChainArgs {
  MapView map;
  ScanView scan;
  int oie;
  int seq;     // 1: the reference's beam-order sum instead of the canonical tree (SLAMHIP_SUM_SEQUENTIAL)
  int verify;  // 1: checked default mode
  McChainCtl *ctl;
  const McPair *tape;  // the window of the matcher's pair tape this match can consume, in HBM
  int n_slots;         // candidates speculated on per super-step: the grid is n_slots + 1 workgroups
  double init[3];
  double td0, rd0;
  unsigned max_failed, max_poses;
  unsigned epoch;
  McHostOut *host;
  McTraceEntry *trace;  // pinned; null = no observer
  int trace_cap;
  // the co-resident form (mc_resident.hip): ONE launch per match, scores exchanged through granules
  struct McResidentCtl *rctl;
  unsigned spin_limit;  // polls of one sweep before the chain gives up
  unsigned tag_epoch;  // co-resident launches on `rctl` so far (hc_tag: NOT the match epoch, which other forms bump too)
  int lds_consts;  // range, cosine, sine of the beams behind a thread's first one are kept in LDS
  int debug_mute;  // testing: workgroup debug_mute - 1 leaves at once, as if it had never become resident
  long long *stamps;  // debugging (slamhip_matcher_debug_stamps): wall-clock stamps of workgroup 1, eight per super-step
};

// the granule block of a co-resident Monte-Carlo chain (HcResidentCtl's layout with this chain's row length)
struct McResidentCtl {
  HcGranule gran[2][kMcSlots + 1];  // super-step k's scores at [k & 1]; the bookkeeping workgroup's at [kMcSlots]
  HcGranule seq[2][kMcSlots + 1];   // beam-order sums of a re-scored super-step
  unsigned fail_epoch;              // = epoch of a match whose workgroups gave up (a bounded spin ran out)
  unsigned pad[3];
};

// mc_resident.hip: the whole match as one launch of a.n_slots + 1 co-resident workgroups; the grid must not exceed
// mc_resident_capacity.  McHostOut::error 4: a workgroup was not resident (bounded spin ran out), 5: more super-steps
// than a tag counts -- nothing has been reported then, the caller runs the chain of kernels
hipError_t launch_mc_chain_resident(const McChainArgs &a, int cell_model, int nt, hipStream_t stream,
                                    hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t mc_resident_capacity(int cell_model, int nt, int n_beams, bool lds_consts, int *out_wgs, int *out_per_cu = nullptr);

// threads per workgroup: 512 or 1024
hipError_t launch_mc_chain_step(const McChainArgs &a, int cell_model, int k, int nt, hipStream_t stream,
                                hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

}  // namespace slamhip
