#!/bin/bash
# tools/kstats_py.sh <script.py> [args...] -- on the GPU box: rocprofv3 kernel stats of one python script, top kernels printed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kstats
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -o ks -- python3 "$@" > gpurun_out/kstats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/kstats/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:18]:
    print("%-100s %6s %9.2f us  %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
tail -3 gpurun_out/kstats.log | cut -c1-300
