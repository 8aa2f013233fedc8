// hc_chain_device.h -- what the chain kernel (hc_chain.hip) and its host driver (matchers.cpp) share.
#pragma once

#include "hc_chain.h"
#include "slamhip_internal.h"

namespace slamhip {

// one scorer call of the walked path, in the reference's order (what GridScanMatcherObserver sees)
struct HcTraceEntry {
  double x, y, theta, score;
  int accepted, pad;
};

// pinned, host-coherent: written by workgroup 0, read by the spinning host
struct HcHostOut {
  double pose[3];
  double best_prob;
  long long calls, evaluated;
  int steps;
  long long rescored;  // super-steps the checked default mode scored a second time, in beam order
  int gm_cx, gm_cy;    // GMapping OOPE: the cache entry after the last scorer call
  double gm_prob;
  int error;           // 1: replay found no terminal round (bug), 2: trace buffer too small, 3: a pose whose whole
                       // scan is one run sat on the path (its cache hand-over needs the sequential replay)
  unsigned progress;   // super-steps started so far in this process_scan
  unsigned done_seq;   // = epoch of the process_scan whose result is above
  // GMapping OOPE: side outputs and raw score of the INITIAL pose (the filter's cross-particle cache check)
  GmPoseInfo first_info;
  double first_raw;
  long long tail_calls;  // co-resident 1-cell / window form: scorer calls of `calls` reported in closed form (the tail
                         // behind an inert or certified root), i.e. not scored
};

// device memory of one matcher
struct HcChainCtl {
  HcState state[2];               // root state of super-step k at [k & 1]
  HcInst walk[2][kHcMaxInst];     // the round instances of super-step k's tree, same parity
  double scores[2][kHcSlots + 7];
  double scores_seq[2][kHcSlots + 7];        // beam-order sums of a re-scored super-step
  unsigned long long hashes[2][kHcSlots + 7];  // term-vector hashes (checked default mode)
  GmPoseInfo infos[2][kHcSlots + 7];  // GMapping OOPE: side outputs of every scored pose
  unsigned done_epoch;            // epoch of the last process_scan that ran to its end
  GmPoseInfo first_info;          // (see HcHostOut)
  double first_raw;
};

// ---- the co-resident form of the chain (hc_resident.hip): one launch per match, scores exchanged inside it
// {score, term-vector fingerprint, tag}: ONE naturally aligned 16-byte write-through store, its own flag (the tag is
// the last dword)
struct alignas(16) HcGranule {
  double score;
  unsigned hash, tag;
};
struct HcResidentCtl {
  HcGranule gran[2][kHcSlots + 7];  // super-step k's scores at [k & 1]
  HcGranule seq[2][kHcSlots + 7];   // beam-order sums of a re-scored super-step
  unsigned fail_epoch;              // = epoch of a match whose workgroups gave up (a bounded spin ran out)
  unsigned pad[3];
};

// ... and over the GMapping OOPE (hc_resident_gm.hip): a pose hands the replay its score AND its cache side outputs
// (GmPoseInfo), four granules of {12 bytes, tag} per slot
struct HcResidentGmCtl {
  HcGranule gran[2][kHcSlots + 7][4];
  unsigned fail_epoch;
  unsigned pad[3];
};

// one match of a BATCH of independent matches (slamhip_matcher_process_scan_batch): its own map and its own scan
struct HcJobView {
  MapView map;
  ScanView scan;
};

struct HcChainArgs {
  MapView map;
  ScanView scan;
  // Independent matches in shared launches (grid.y = match; PoseEnumerationScanMatcher::process_scan once per robot,
  // pose_enumeration_scan_matcher.h:31-77): chain c reads map and scan from jobs[c] instead of the two views above
  // (scan.n above is then the largest beam count of the batch: the launch's LDS size).  null: one map, one scan.
  const HcJobView *jobs;
  int oie;
  GmParams gm;                 // GMapping OOPE (cell model GMAPPING): threshold and window
  int gm_cx, gm_cy;            // ... and the cache entry the match starts from (-1 = empty)
  double gm_prob;
  int seq;  // 1: the reference's beam-order sum instead of the canonical tree (SLAMHIP_SUM_SEQUENTIAL)
  int verify;  // 1 (default mode, point OOPE): comparisons too close for the tree sum to settle are re-decided from
               // beam-order sums (hc_round_decide)
  // Several chains in one launch (the GMapping filter: one chain per particle, same map, same scan): grid.y = chain,
  // `ctl` and `host` are arrays, `inits` holds the chains' initial poses (null: one chain, `init` below) and every
  // chain that ends bumps *n_done.
  HcChainCtl *ctl;
  const double *inits;
  unsigned *n_done;
  // ... each on its own copy-on-write map: `map` describes the tile pool, chain c gathers through the tile table of
  // slot slots[c] (null: the one dense map)
  const int *tables;
  const int *slots;
  int table_stride;
  const HcShape *shapes;  // kHcShapes of them
  int max_inst;  // instances of the largest shape: the grid is 6 x max_inst + 1 workgroups
  unsigned long long n_inst;  // round instances of shape b in byte b (a dynamic index into an array of
                              // kernel arguments is a global load: 1 us on the replay's critical path)
  double init[3];
  double dt0, dr0;
  unsigned max_failed;
  int shape0;
  unsigned epoch;
  HcHostOut *host;
  HcTraceEntry *trace;  // pinned; null = no observer.  Chain c writes at trace + c * trace_stride, at most
  int trace_cap;        // trace_cap entries
  int trace_stride;
  long long *stamps;    // debugging: 8 wall-clock stamps (100 MHz) per super-step of workgroup 1, or null
  // co-resident form only (hc_resident.hip)
  HcResidentCtl *rctl;   // one per chain
  HcResidentGmCtl *rctl_gm;  // ... of the GMapping form
  unsigned spin_limit;   // polls of one sweep before the chain gives up (the host sizes it from the matches it has seen)
  unsigned tag_epoch;    // co-resident launches on these blocks so far: the epoch bits of a granule's tag (hc_tag) -- NOT
                         // `epoch`, which the other forms of the chain bump too
  unsigned *h_all_done;  // pinned; a batch's last chain to end stores the epoch here (null: a lone chain)
  int debug_mute;        // testing: workgroup debug_mute - 1 leaves at once, as if it had never become resident
  int lds_consts;  // co-resident 1-cell form: range, cosine, sine of the beams behind a thread's first one are kept in LDS
  int pair;              // co-resident batch form: a workgroup (512 threads) scores two poses per super-step
  int tab_offset;        // co-resident 1-cell / window form: where the table of next poses starts in dynamic LDS, in doubles
                         // (set by the launcher: hc_resident_tab_offset)
  int oope;              // SLAMHIP_OOPE_OBSTACLE (0), or a window OOPE (max / mean / overlap) with its analysis area
  double area[4];
  int inert_tail;        // co-resident 1-cell / window form (SLAMHIP_OPT_INERT_TAIL): >= 1: a root whose rounds only repeat
                         // its own pose (hc_inert) ends the chain in closed form instead of being scored
                         // 6 x (limit - failed) + 1 more times; 2: so does a root the bookkeeping workgroup has CERTIFIED
                         // for the next steps (1-cell form; hc_resident.hip "certificate")
};

// threads per workgroup: 256, 512 or 1024; n_chains > 1: the multi-chain form (see HcChainArgs::inits)
hipError_t launch_hc_chain_step(const HcChainArgs &a, int cell_model, int k, int nt, hipStream_t stream,
                                hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int n_chains = 1);
// the whole match in ONE launch of co-resident workgroups (1-cell OOPE); the caller has checked the grid against
// hc_resident_capacity.  HcHostOut::error 4: a workgroup was not resident (bounded spin ran out), 5: more super-steps
// than a tag counts -- either way nothing was reported, the kernel chain redoes the match
hipError_t launch_hc_chain_resident(const HcChainArgs &a, int cell_model, int nt, hipStream_t stream,
                                    hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int n_chains = 1);
hipError_t hc_resident_capacity(int cell_model, int nt, bool batch, bool window, int n_beams, bool lds_consts, int max_inst,
                                int *out_wgs, int *out_per_cu = nullptr, bool pair = false);
size_t hc_resident_lds_bytes(int nt, int n_beams, bool lds_consts, int max_inst, bool pair = false);
// the GMapping OOPE's co-resident form (hc_resident_gm.hip): one chain, or n_chains of a filter step (grid.y = chain:
// HcChainArgs::inits / n_done / h_all_done / tables / slots as in launch_hc_chain_step)
hipError_t launch_hc_chain_resident_gm(const HcChainArgs &a, int nt, hipStream_t stream, hipEvent_t ev_start = nullptr,
                                       hipEvent_t ev_stop = nullptr, int n_chains = 1);
hipError_t hc_resident_gm_capacity(int nt, int n_beams, int *out_wgs, int *out_per_cu = nullptr);
// one thread: copies the number of finished chains to pinned memory and publishes a launch number (the host's
// view of a burst of multi-chain super-steps)
hipError_t launch_chain_marker(const unsigned *n_done, unsigned *h_done_count, unsigned *flag, unsigned seq,
                               hipStream_t stream);

}  // namespace slamhip
