// Host-side sanitizer check of the speculation builders (matchers.h): built with -fsanitize=address
// and run on the CPU by tests/test_host_sanitizers.py.  No device code is executed.
// Regression: build_rounds read the parent round's enumerator through a pointer into rounds_ after
// rounds_.emplace_back() had reallocated the vector (r01: flaky GMapping golden).
#include <cstdio>
#include <random>

#include "matchers.h"

using namespace slamhip;

int main() {
  std::mt19937 rng(7);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  long long evals = 0, rounds = 0;
  for (int rep = 0; rep < 200; ++rep) {
    SpecTree tree;  // fresh vectors every time: every growth step of rounds_ is exercised
    tree.min_reach = rep % 2 ? 0.3 : 0.002;
    HillClimbingPoseEnumerator pe(6 + rep % 5, 0.1, 0.1);
    Pose best{0.1 * rep, -0.2, 0.3};
    while (pe.has_next()) {
      tree.build(pe, best, 126 + 6 * (rep % 7), 0.004 + 0.45 * u(rng));
      if (tree.evals.empty()) break;
      evals += (long long)tree.evals.size();
      // replay with random outcomes exactly like MatchJob::consume
      int node = tree.root;
      while (node >= 0) {
        const SpecTree::Node &nd = tree.nodes[node];
        const Pose c = pe.next(best);
        if (std::memcmp(&c, &tree.evals[nd.eval], sizeof(Pose)) != 0) {
          std::printf("speculated candidate differs from the enumerator's\n");
          return 1;
        }
        const bool ok = u(rng) < 0.1;
        pe.feedback(ok);
        if (ok) best = c;
        node = nd.child[ok ? 1 : 0];
        ++rounds;
      }
      if (node == SpecTree::kEnd) break;
    }
  }
  // Monte-Carlo and brute-force chains
  for (int rep = 0; rep < 20; ++rep) {
    SpecTree tree;
    GaussianPoseEnumerator mc(1234 + rep, 0.2, 0.1, 20, 300);
    Pose best{0, 0, 0};
    while (mc.has_next()) {
      tree.build(mc, best, 64 + rep, 0.05);
      if (tree.evals.empty()) break;
      int node = tree.root;
      while (node >= 0) {
        const SpecTree::Node &nd = tree.nodes[node];
        const Pose c = mc.next(best);
        if (std::memcmp(&c, &tree.evals[nd.eval], sizeof(Pose)) != 0) return 2;
        const bool ok = u(rng) < 0.05;
        mc.feedback(ok);
        if (ok) best = c;
        node = nd.child[ok ? 1 : 0];
      }
      if (node == SpecTree::kEnd) break;
    }
  }
  std::printf("ok %lld evals %lld walked\n", evals, rounds);
  return 0;
}
