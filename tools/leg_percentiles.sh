#!/bin/bash
# tools/leg_percentiles.sh <leg> <kernel-substring> [...] -- rocprofv3 kernel trace of one bench.py leg on the GPU box,
# duration percentiles (us) of the named kernels (a mean hides a tail: k_mu_lines' p50 4 us, p90 80 us)
LEG=$1; shift
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/lp
rocprofv3 --kernel-trace --output-format csv -d /tmp/lp -o lp -- python3 $GRAFT_REPO_ROOT/bench.py --legs $LEG --steps 3 --warmup 1 --no-cpu > /tmp/lp.log 2>&1
python3 - "$@" <<'PY'
import csv, glob, sys
import numpy as np
f = glob.glob('/tmp/lp/**/*kernel_trace.csv', recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    for k in sys.argv[1:]:
        if k in r['Kernel_Name']:
            d.setdefault(k, []).append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    v = np.array(v)
    print(k, len(v), 'p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f mean %.1f' % (*np.percentile(v, [10, 50, 90, 99]), v.max(), v.mean()))
PY
grep '^{"metric' /tmp/lp.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
pf=d.get('particle_filter',{})
for k in ('with_map_update','with_particle_maps'):
    if k in pf: print(k, pf[k].get('ms_per_step'), pf[k].get('value'))
if 'world_loop' in d: print('world', d['world_loop'].get('ms_per_scan'), d['world_loop'].get('value'))
"
