#!/bin/bash
# GPU box: bench.py on several workloads.  usage: gpu_bench_cfgs.sh TAG "args for run 1" "args for run 2" ...
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=$1; shift
i=0
for A in "$@"; do
  i=$((i+1))
  timeout -k 10 900 python bench.py $A > gpurun_out/bench_${TAG}_$i.json 2> gpurun_out/bench_${TAG}_$i.err || { tail -30 gpurun_out/bench_${TAG}_$i.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}_$i.json"))
print("[$A]", "value=%.4g pairs/s"%d["value"], "ms/step=%.3f"%d["ms_per_step"], {k: round(v,4) for k,v in d["kernel_ms"].items()}, "sel", d["config"]["selected_pairs"], "surv", d["config"]["stage1_survivors"], "roof %.3g"%d["roofline"]["frac"], "cpu", d.get("cpu_baseline",{}).get("value"))
PY
done
