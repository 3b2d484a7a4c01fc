#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of bench.py for one algorithm  (usage: gpu_prof.sh TAG ALGO [extra bench args])
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; ALGO=$2; shift 2
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o trace -- python3 $R/bench.py --steps 10 --warmup 2 --algo $ALGO --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.log 2>&1 || { tail -30 $R/gpurun_out/prof_$TAG.log; exit 1; }
cd $R
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_$TAG/trace_kernel_stats.csv")):
    print(r["Name"][:70].ljust(72), r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3), r["Percentage"])
PY
tail -1 gpurun_out/prof_$TAG.log | cut -c1-300
