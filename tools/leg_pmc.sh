#!/bin/bash
# tools/leg_pmc.sh <leg> <kernel-substring> <counter> [...] -- one rocprofv3 --pmc pass over one bench.py leg on the GPU
# box, per-kernel means (tools/pmc_by_kernel.py)
LEG=$1; KER=$2; shift 2
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/lpm
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/lpm -o pm -- python3 $GRAFT_REPO_ROOT/bench.py --legs $LEG --steps 2 --warmup 1 --no-cpu > /tmp/lpm.log 2>&1
f=$(find /tmp/lpm -name '*counter_collection.csv' | head -1)
PMC_MIN_N=2 python3 $GRAFT_REPO_ROOT/tools/pmc_by_kernel.py $f "$KER"
