// bf_device.h -- what bf_device.hip and its host driver (matchers.cpp) share.
#pragma once

#include <hip/hip_runtime.h>

namespace slamhip {

struct BfPoseArgs {
  double init[3];     // pose 0: the match's initial pose, scored first
  double base[3];     // the pose the enumerator latched at its FIRST next() ever -- the first match's initial pose:
                      // reset() does not clear it (brute_force_scan_matcher.h:27-40), every later match of the
                      // same matcher enumerates around that pose
  const double *off;  // nx x-offsets, ny y-offsets, nt theta-offsets, as the enumerator accumulates them
  int nx, ny, nt;
  long long n;        // 1 + nx ny nt poses: the initial pose, then x fastest, y, theta
  double *poses;      // n x 3
};

// pinned, host-coherent
struct BfHostOut {
  long long best_index;  // 0 = the initial pose stays the best
  double best_score;
  long long accepts;     // acceptances of the walk (on_pose_update events behind the initial one)
  int ambiguous;         // checked default mode: a comparison of the walk that the tree sums cannot settle
  unsigned seq;
};

constexpr int kBfMaxBlocks = 1 << 16;  // blocks of 1024 candidates (2^26 poses)

struct BfArgmaxArgs {
  const double *scores;               // n
  const unsigned long long *fprints;  // n, or null
  long long n;
  int verify;
  BfHostOut *out;
  unsigned seq;
  // scratch: per candidate the best candidate in front of it inside its block, per block its own best, three words
  // (accepts, ambiguous, blocks done -- zero between matches)
  long long *pidx;
  double *agg_s;
  long long *agg_i;
  unsigned *counters;
};

hipError_t launch_bf_poses(const BfPoseArgs &a, hipStream_t stream);
hipError_t launch_bf_argmax(const BfArgmaxArgs &a, hipStream_t stream);

}  // namespace slamhip
