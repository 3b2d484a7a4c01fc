#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, bench, rocprofv3 kernel-trace stats, PMC passes (own runs).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
ALGO=${2:-auto}
python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1 || { tail -30 gpurun_out/smoke_$TAG.log; exit 1; }
tail -1 gpurun_out/smoke_$TAG.log
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --algo $ALGO > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -30 gpurun_out/bench_$TAG.err; exit 1; }
cat gpurun_out/bench_$TAG.json
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o trace -- python3 $R/bench.py --steps 10 --warmup 2 --algo $ALGO --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1 || { tail -30 $R/gpurun_out/prof_$TAG.log; exit 1; }
cd $R
find gpurun_out/prof_$TAG -name "*stats*" | head
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${c}_$TAG -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --algo $ALGO --no-cpu-baseline > $R/gpurun_out/pmc_${c}_$TAG.log 2>&1 || { tail -30 $R/gpurun_out/pmc_${c}_$TAG.log; exit 1; }
  cd $R
done
find gpurun_out -name "*.csv" | head -20
