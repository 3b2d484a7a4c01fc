#!/bin/bash
# 16-byte loads in flight per thread of the signature build: 4 (the library) / 8 / 16 (SELHIP_LIB builds)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
for W in "--workload cfg3" "--workload cfg3 --genomes 28280" "--workload cfg4"; do
  for L in 4 8 16; do
    T=$(echo "$W" | tr -d ' -')_L$L
    if [ $L = 4 ]; then unset SELHIP_LIB; else export SELHIP_LIB=scripts/microbench/libselhip_sig$L.so; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras $W > $O/p_$T.json 2> $O/p_$T.err || { tail -20 $O/p_$T.err; exit 1; }
    python - <<PY
import json
d=json.load(open("$O/p_$T.json"))
print("%-36s loads=%2d ms/step=%.4f sigbuild=%.4f" % ("$W", $L, d["ms_per_step"], d["kernel_ms"]["sigbuild"]))
PY
  done
done
