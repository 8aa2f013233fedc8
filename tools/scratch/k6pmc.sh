#!/bin/bash
# only the K6 PMC passes of tools/profile.sh (into the same gpurun_out/r02 tree)
set -u
TAG=r02
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
pmc() { local wl=$1 pass=$2; shift 2; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $OUT/pmc_${wl}_$pass -o pmc -- python3 $ROOT/bench.py "$@" > $OUT/pmc_${wl}_$pass.log 2>&1; }
for c in FETCH_SIZE WRITE_SIZE; do
  pmc pf_update $c $c -- --legs pf_update --steps 3 --warmup 1 --no-cpu
  pmc pf_maps $c $c -- --legs pf_maps --steps 3 --warmup 1 --no-cpu
  pmc cfg5 $c $c -- --legs cfg5 --steps 3 --warmup 1 --no-cpu
done
ls $OUT | grep pmc_ | head -30
