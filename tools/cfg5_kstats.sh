cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -o p5 -- python3 $GRAFT_REPO_ROOT/bench.py --legs cfg5 --steps 3 --warmup 1 --no-cpu > /tmp/p5.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/p5/**/p5_kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
