#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python3 $ROOT/bench.py --detail-out $OUT/$name.bench.json "$@" > $OUT/$name.bench.log 2>&1
  python3 $ROOT/bench.py --detail-out $OUT/$name.plain.json "$@" 2> /dev/null | grep '^{"metric' | tail -1 > $OUT/$name.line.json
}
python3 $ROOT/bench.py --steps 20 --warmup 5 --detail-out $OUT/default.plain.json > $OUT/default.stdout 2> $OUT/default.err
tail -n 1 $OUT/default.stdout > $OUT/default.line.json
run world --legs world --steps 3 --warmup 1 --no-cpu
run world_viny --legs world_viny --steps 3 --warmup 1 --no-cpu
python3 -c "
import json; d=json.load(open('$OUT/default.line.json')); print(d['ms_per_step'], d['config']['ms_per_step_every_call_scored'], d['roofline']['avg_launch_us'], d['legs']['world_loop']['ms_per_step'], d['legs']['world_loop_viny']['ms_per_step'])"
