#!/bin/bash
# genomes per tile of the tiled signature build (8 / 16 / 32) at three sizes: the build's own time and the step
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_pass or synthetic_vs_oracle or signature_cache" > $O/pytest_o.log 2>&1 || { tail -40 $O/pytest_o.log; exit 1; }
tail -1 $O/pytest_o.log
for W in "--workload cfg3" "--workload cfg3 --genomes 28280" "--workload cfg4"; do
  for G in 8 16 32; do
    T=$(echo "$W" | tr -d ' -')_$G
    timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras $W --param sig_tile_g=$G > $O/o_$T.json 2> $O/o_$T.err || { tail -20 $O/o_$T.err; exit 1; }
    python - <<PY
import json
d=json.load(open("$O/o_$T.json"))
print("%-36s tile=%2d ms/step=%.4f sigbuild=%.4f" % ("$W", $G, d["ms_per_step"], d["kernel_ms"]["sigbuild"]))
PY
  done
done
