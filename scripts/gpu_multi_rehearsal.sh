#!/bin/bash
# GPU box (1 GPU): rehearse the N>1 bench path with 2 and 4 ranks sharing the card; collectives over gloo.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
for N in 2 4; do
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) \
     bench.py --gpus $N --steps 5 --warmup 2 --backend gloo > gpurun_out/bench_gloo_$N.json 2> gpurun_out/bench_gloo_$N.err || { tail -40 gpurun_out/bench_gloo_$N.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_gloo_$N.json").read().strip().splitlines()[-1])
print("ranks=$N", "value=%.4g"%d["value"], "ms/step=%.3f"%d["ms_per_step"], d["config"]["n_genomes"], d["config"]["pairs_per_step"], d["config"]["selected_pairs"], d["scaling"])
PY
done
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single', '%.4g'%d['value'], '%.3f'%d['ms_per_step'], d['roofline']['traffic'])"
