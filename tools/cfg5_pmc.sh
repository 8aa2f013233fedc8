#!/bin/bash
# tools/cfg5_pmc.sh <counter> [<counter> ...] -- one rocprofv3 --pmc pass over the cfg5 leg, per-kernel means of k_mu_*
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm5
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pm5 -o pm -- python3 $GRAFT_REPO_ROOT/bench.py --legs cfg5 --steps 2 --warmup 1 --no-cpu > /tmp/pm5.log 2>&1
f=$(find /tmp/pm5 -name '*counter_collection.csv' | head -1)
PMC_MIN_N=2 python3 $GRAFT_REPO_ROOT/tools/pmc_by_kernel.py $f k_mu_classify k_mu_emit
