#!/bin/bash
for cfg in "1 1024" "2 1024" "2 512" "4 512" "4 1024" "8 512"; do
  set -- $cfg
  timeout 300 python bench.py --gpus $1 --backend gloo --legs none --no-cpu --chain $2 --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ranks $1 threads $2:', round(d['value']/1e9,2), 'G units/s', round(d['ms_per_step'],4), 'ms/step')"
done
