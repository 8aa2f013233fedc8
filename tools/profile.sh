#!/bin/bash
# tools/profile.sh <round-tag> -- run on the GPU box (gpurun): rocprofv3 kernel-trace + stats of the
# default bench command (hc), of the sweep and of mc, then PMC passes for hc and sweep (FETCH_SIZE /
# WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes; never mixed with trace domains
# other than --kernel-trace).  Output: gpurun_out/<tag>/; tools/summarize_profiles.py <tag> then
# copies the summaries into profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() { # name, extra bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 $ROOT/bench.py "$@" > $OUT/$name.bench.log 2>&1
  grep '^{"metric' $OUT/$name.bench.log | tail -1 > $OUT/$name.bench.json
  # the same command without the profiler: under rocprofv3 the HIP events attached to a dispatch read
  # ~8 us long (17 vs 9 us), so the live kernel time to compare with the trace is the un-profiled one
  python3 $ROOT/bench.py "$@" 2> /dev/null | grep '^{"metric' | tail -1 > $OUT/$name.plain.json
}
pmc() { # workload-name, pass-name, counters..., then "--", bench args
  local wl=$1 pass=$2; shift 2
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $OUT/pmc_${wl}_$pass -o pmc -- python3 $ROOT/bench.py "$@" > $OUT/pmc_${wl}_$pass.log 2>&1
}
run hc --steps 50 --warmup 5 --cpu-seconds 6
run sweep --workload sweep --steps 200 --warmup 10 --no-cpu
run mc --workload mc --steps 20 --warmup 3 --no-cpu
for c in FETCH_SIZE WRITE_SIZE; do
  pmc sweep $c $c -- --workload sweep --steps 20 --warmup 2 --no-cpu
  pmc hc $c $c -- --steps 10 --warmup 2 --no-cpu --no-pf
  pmc pf $c $c -- --steps 5 --warmup 1 --no-cpu --pf-steps 4   # k_score_gmapping of the particle-filter leg
done
pmc sweep sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- --workload sweep --steps 20 --warmup 2 --no-cpu
find $OUT -name "*.csv" | head -60
