#!/bin/bash
# GPU box, one call: full gpu test-suite, smoke, sketch-build bench, default bench (+ optional extra bench args as $2..)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${1:-full}; shift || true
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_$TAG.log 2>&1 || { tail -60 gpurun_out/pytest_gpu_$TAG.log; exit 1; }
tail -2 gpurun_out/pytest_gpu_$TAG.log
python __graft_entry__.py smoke 2>&1 | tail -1
python scripts/bench_build.py > gpurun_out/bench_build_$TAG.json 2> gpurun_out/bench_build_$TAG.err || { tail -20 gpurun_out/bench_build_$TAG.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/bench_build_$TAG.json')); print('build: gpu %.3g kmers/s (%.2f ms), ref cpu %s' % (d['gpu_kmers_per_s'], d['gpu_ms'], d.get('cpu_reference')))"
i=0
for A in "--steps 20 --warmup 3 --no-cpu-baseline" "$@"; do
  i=$((i+1))
  timeout -k 10 900 python bench.py $A > gpurun_out/bench_${TAG}_$i.json 2> gpurun_out/bench_${TAG}_$i.err || { tail -30 gpurun_out/bench_${TAG}_$i.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}_$i.json"))
print("[$A]", "value=%.4g pairs/s"%d["value"], "ms/step=%.3f"%d["ms_per_step"], {k: round(v,4) for k,v in d["kernel_ms"].items()}, "sel", d["config"]["selected_pairs"])
PY
done
