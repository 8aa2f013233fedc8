#!/bin/bash
for cfg in "1024 40" "1024 36" "1024 28" "512 32" "512 24"; do
  set -- $cfg
  SLAMHIP_GM_CHAIN_THREADS=$1 SLAMHIP_GM_CHAIN_INST=$2 timeout 300 python bench.py --legs pf_update --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); w=d['particle_filter']['with_map_update']; print('gm $cfg', round(w['value']), round(w['ms_per_step'],2))"
done
for cfg in "512 64" "512 42" "1024 42" "1024 32" "256 64" "512 52"; do
  set -- $cfg
  SLAMHIP_HC_CHAIN_INST=$2 timeout 300 python bench.py --legs none --no-cpu --chain $1 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('hc $cfg', round(d['ms_per_step'],4), d['config']['launches_per_step'], round(d['roofline']['avg_launch_us'],2))"
done
