#!/bin/bash
# scratch: the per-particle-maps step at 13 and 100 particles, device chains against lock-step jobs
for n in 13 100; do for c in 1 0; do
  SLAMHIP_PF_CHAIN=$c timeout 300 python bench.py --legs pf_maps --no-cpu --steps 10 --particles $n 2>/dev/null | tail -1 > /tmp/x.json
  python3 -c "
import json;d=json.load(open('/tmp/x.json'));v=d['particle_filter']['with_particle_maps'];print('particles',$n,'chain',$c,round(v['ms_per_step'],3),'ms/step')"
done; done
