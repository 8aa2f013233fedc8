#!/bin/bash
run() { timeout 300 python bench.py --legs pf --no-cpu --particles $2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); w=d['particle_filter']; print('$1 n=$2', round(w['value']), round(w['ms_per_step'],3), w['launches_last_step'], round(w['roofline']['avg_launch_us'],1))"; }
for n in 100 25 13; do
  for w in 200 256 300 500; do SLAMHIP_PF_CHAIN_WGS=$w run "chains $w" $n; done
done
