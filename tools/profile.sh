#!/bin/bash
# tools/profile.sh <round-tag> -- run on the GPU box (gpurun): rocprofv3 kernel-trace + stats of the
# default bench command and of the sweep, then PMC passes (FETCH_SIZE / WRITE_SIZE separately, as
# MI355X_MICROARCH.md prescribes).  Summaries land in gpurun_out/<tag>/; copy the CSV/JSON
# summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() { # name, extra bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 $ROOT/bench.py "$@" > $OUT/$name.bench.log 2>&1
  grep "^{\"metric" $OUT/$name.bench.log | tail -1 > $OUT/$name.bench.json
}
run hc --steps 50 --warmup 5 --cpu-seconds 6
run sweep --workload sweep --steps 200 --warmup 10 --no-cpu
run mc --workload mc --steps 20 --warmup 3 --no-cpu
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python3 $ROOT/bench.py --workload sweep --steps 20 --warmup 2 --no-cpu > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $ROOT/bench.py --workload sweep --steps 20 --warmup 2 --no-cpu > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -40
