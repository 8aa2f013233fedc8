#!/bin/bash
# tools/profile.sh <round-tag> -- run on the GPU box (gpurun).  One rocprofv3 --kernel-trace --stats run PER LEG of
# bench.py (so that every kernel's time can be re-derived from profiles/ alone), the same commands without the
# profiler (HIP-event kernel times to compare with), and PMC passes -- FETCH_SIZE / WRITE_SIZE / SQ counters, each
# in its own pass with --kernel-trace only, as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/<tag>/;
# tools/summarize_profiles.py <tag> then copies the summaries into profiles/.
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() { # name, bench args...
  local name=$1; shift
  # (r06: stdout carries a digest line <= 8 KB, the full record is the --detail-out sidecar: that is what profiles/ keeps)
  # (--no-tail-ab: the headline's every-call-scored comparison is not run under the profiler -- its 72 us launches would be
  # averaged into the default path's in the kernel stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 $ROOT/bench.py --no-tail-ab --detail-out $OUT/$name.bench.json "$@" > $OUT/$name.bench.log 2>&1
  python3 $ROOT/bench.py --detail-out $OUT/$name.plain.json "$@" 2> /dev/null | grep '^{"metric' | tail -1 > $OUT/$name.line.json
}
pmc() { # workload-name, pass-name, counters..., then "--", bench args
  local wl=$1 pass=$2; shift 2
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $OUT/pmc_${wl}_$pass -o pmc -- python3 $ROOT/bench.py --no-tail-ab --detail-out $OUT/pmc_${wl}_$pass.detail.json "$@" > $OUT/pmc_${wl}_$pass.log 2>&1
}
# 1. PMC passes first: their summary (HBM bytes and VALU instructions per launch) is what the bench lines of
#    step 2 quote as roofline.traffic / roofline_valu, so it has to exist -- in THIS copy of the repo -- before them
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for c in FETCH_SIZE WRITE_SIZE; do
  pmc hc $c $c -- --legs none --steps 10 --warmup 2 --no-cpu
  pmc sweep $c $c -- --workload sweep --steps 20 --warmup 2 --no-cpu
  pmc mc $c $c -- --workload mc --legs none --steps 5 --warmup 1 --no-cpu
  pmc pf $c $c -- --legs pf --steps 3 --warmup 1 --no-cpu --pf-steps 4
done
# the map-update pipeline (K6) of the three legs that run it: HBM bytes per pipeline = all k_mu_* / rocprim dispatches
for c in FETCH_SIZE WRITE_SIZE; do
  pmc pf_update $c $c -- --legs pf_update --steps 3 --warmup 1 --no-cpu
  pmc pf_maps $c $c -- --legs pf_maps --steps 3 --warmup 1 --no-cpu
  pmc cfg5 $c $c -- --legs cfg5 --steps 3 --warmup 1 --no-cpu
  pmc world $c $c -- --legs world --steps 3 --warmup 1 --no-cpu
done
pmc hc sq $SQ -- --legs none --steps 10 --warmup 2 --no-cpu
pmc sweep sq $SQ -- --workload sweep --steps 20 --warmup 2 --no-cpu
pmc mc sq $SQ -- --workload mc --legs none --steps 5 --warmup 1 --no-cpu
pmc pf sq $SQ -- --legs pf --steps 3 --warmup 1 --no-cpu --pf-steps 4
python3 $ROOT/tools/summarize_profiles.py $TAG > /dev/null 2>&1   # writes profiles/${TAG}_traffic.json here
# 2. kernel traces and the un-profiled lines
# the driver's command (all legs, CPU baselines): un-profiled only -- this is the line BENCH_rNN will hold
# (stdout kept whole: default.stdout's LAST line is what the driver parses -- default.line.json; the sidecar is the record)
python3 $ROOT/bench.py --steps 20 --warmup 5 --detail-out $OUT/default.plain.json > $OUT/default.stdout 2> $OUT/default.err
tail -n 1 $OUT/default.stdout > $OUT/default.line.json
run hc --legs none --steps 50 --warmup 5 --no-cpu
run sweep --workload sweep --steps 200 --warmup 10 --no-cpu
run mc --workload mc --legs none --steps 20 --warmup 3 --no-cpu
run pf --legs pf --steps 3 --warmup 1 --no-cpu
run pf_update --legs pf_update --steps 3 --warmup 1 --no-cpu
run pf_maps --legs pf_maps --steps 3 --warmup 1 --no-cpu
run cfg5 --legs cfg5 --steps 3 --warmup 1 --no-cpu
run world --legs world --steps 3 --warmup 1 --no-cpu
run replicas --legs replicas --steps 3 --warmup 1 --no-cpu
run bf --legs bf --steps 3 --warmup 1 --no-cpu
run mc_leg --legs mc --steps 3 --warmup 1 --no-cpu
run world_viny --legs world_viny --steps 3 --warmup 1 --no-cpu
python3 $ROOT/tools/hc_batch_stamps.py 2> /dev/null > $OUT/batch_stamps.txt
python3 $ROOT/tools/hc_chain_stamps.py > $OUT/chain_stamps.txt 2>&1
python3 $ROOT/tools/hc_resident_stamps.py 2> /dev/null > $OUT/resident_stamps.txt
ls $OUT | head -80
