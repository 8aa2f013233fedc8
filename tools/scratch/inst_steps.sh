#!/bin/bash
# super-steps per GMapping HC(6) match as a function of the tree size (kernels launched / matches, run-ahead included)
for inst in 1 2 4 8 42; do
  SLAMHIP_GM_CHAIN_INST=$inst tools/kstats.sh --legs pf_update --no-cpu | grep "k_hc_chain_step<2" | awk -v i=$inst '{print "inst", i, "calls", $(NF-3), "avg us", $(NF-2)}'
done
