#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_pipe.log 2>&1 || { tail -60 gpurun_out/pytest_gpu_pipe.log; exit 1; }
tail -2 gpurun_out/pytest_gpu_pipe.log
for P in 0 2 4 8; do
  for W in cfg3 cfg4; do
    python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --pipeline $P > gpurun_out/bench_pipe_${W}_$P.json 2> gpurun_out/bench_pipe_${W}_$P.err || { tail -20 gpurun_out/bench_pipe_${W}_$P.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/bench_pipe_${W}_$P.json')); print('$W pipeline=$P', 'value=%.4g'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms'].items()}, d['config']['selected_pairs'])"
  done
done
