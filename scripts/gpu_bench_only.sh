#!/bin/bash
# GPU box: bench.py for each algorithm given (no tests).  usage: gpu_bench_only.sh TAG algo...
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=$1; shift
for ALGO in "$@"; do
  timeout -k 10 600 python bench.py --steps 20 --warmup 3 --algo $ALGO --no-cpu-baseline > gpurun_out/bench_${TAG}_$ALGO.json 2> gpurun_out/bench_${TAG}_$ALGO.err || { tail -30 gpurun_out/bench_${TAG}_$ALGO.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}_$ALGO.json"))
print("$ALGO", "value=%.4g pairs/s"%d["value"], "ms/step=%.3f"%d["ms_per_step"], {k: round(v,4) for k,v in d["kernel_ms"].items()}, d["config"]["selected_pairs"], d["config"]["stage1_survivors"])
PY
done
