#!/bin/bash
# GPU box: one rocprofv3 --pmc pass (counters in $3...) over bench.py for algo $2; prints per-kernel averages
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; ALGO=$2; shift 2
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$TAG -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --algo $ALGO --no-cpu-baseline > $R/gpurun_out/pmc_$TAG.log 2>&1 || { tail -30 $R/gpurun_out/pmc_$TAG.log; exit 1; }
cd $R
python - <<PY
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("gpurun_out/pmc_$TAG/pmc_counter_collection.csv")):
    agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if any(x in k for x in ("sig_join","smh_stream","hll_union","verify","ertl_select")):
        print(k, {c: "%.4g"%(sum(x)/len(x)) for c,x in v.items()})
PY
