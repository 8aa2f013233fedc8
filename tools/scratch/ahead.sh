#!/bin/bash
# scratch: the headline match, two runs (used to compare builds with another run-ahead depth of the chain)
for i in 1 2 3; do
  timeout 200 python bench.py --legs none --no-cpu --steps 100 2>/dev/null | tail -1 > /tmp/a.json
  python3 -c "
import json;d=json.load(open('/tmp/a.json'));print('hc', round(d['ms_per_step'],4), d['config']['launches_per_step'])"
done
