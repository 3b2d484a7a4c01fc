#!/bin/bash
# GPU box: full gpu test-suite, then bench for the requested algorithms.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${1:-r01}
shift || true
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_$TAG.log 2>&1 || { tail -60 gpurun_out/pytest_gpu_$TAG.log; exit 1; }
tail -3 gpurun_out/pytest_gpu_$TAG.log
for ALGO in "$@"; do
  timeout -k 10 600 python bench.py --steps 20 --warmup 3 --algo $ALGO --no-cpu-baseline > gpurun_out/bench_${TAG}_$ALGO.json 2> gpurun_out/bench_${TAG}_$ALGO.err || { tail -30 gpurun_out/bench_${TAG}_$ALGO.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}_$ALGO.json"))
print("$ALGO", "value=%.4g pairs/s"%d["value"], "ms/step=%.3f"%d["ms_per_step"], d["kernel_ms"], d["config"]["selected_pairs"], d["config"]["stage1_survivors"])
PY
done
