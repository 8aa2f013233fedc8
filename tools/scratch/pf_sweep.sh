#!/bin/bash
for wb in 160 256 320 512; do
  for rep in 1 2; do
  SLAMHIP_K3_WIDE_BELOW=$wb timeout 300 python bench.py --legs pf --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); w=d['particle_filter']; print('wide_below $wb:', round(w['value']), round(w['ms_per_step'],3))"
  done
done
