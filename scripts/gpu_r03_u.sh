#!/bin/bash
# join tile height sweep at cfg3 (and 8 waves per block): does an integral number of wave rounds help?
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
for Q in 16 32 48 64 80 96; do
  for W in 4 8; do
    timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --param join_qt=$Q --param join_wpb=$W > $O/u_${Q}_$W.json 2> $O/u_${Q}_$W.err || { tail -20 $O/u_${Q}_$W.err; exit 1; }
    python - <<PY
import json
d=json.load(open("$O/u_${Q}_$W.json"))
k=d["kernel_ms"]
print("qt=%3d wpb=%d ms/step=%.4f join=%.4f join_in_region=%.4f" % ($Q, $W, d["ms_per_step"], k["join"], k.get("join_in_timed_region", -1)))
PY
  done
done
