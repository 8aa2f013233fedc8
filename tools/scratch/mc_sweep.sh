#!/bin/bash
for cfg in "1024 252" "512 384" "1024 128" "512 252"; do
  set -- $cfg
  SLAMHIP_MC_CHAIN_SLOTS=$2 timeout 200 python bench.py --workload mc --legs none --no-cpu --chain $1 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$cfg', round(d['ms_per_step'],4), c['launches_per_step'], round(d['roofline']['avg_launch_us'],2), c['host_us_last_step'], c['poses_evaluated_per_step'], round(c['kernel_busy_frac'],2))"
done
