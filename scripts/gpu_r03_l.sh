#!/bin/bash
# where do two chunk lanes start to pay?  single GPU, cfg3 data at growing genome counts, pipeline off / 2
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
for N in 28280 33000 38000 45000; do
  for P in 0 2; do
    timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --genomes $N --pipeline $P > $O/l_${N}_$P.json 2> $O/l_${N}_$P.err || { tail -20 $O/l_${N}_$P.err; exit 1; }
    python - <<PY
import json
d=json.load(open("$O/l_${N}_$P.json"))
print("n=$N pipeline=$P ms/step=%.4f pairs=%.3g" % (d["ms_per_step"], d["config"]["pairs_per_step"]))
PY
  done
done
