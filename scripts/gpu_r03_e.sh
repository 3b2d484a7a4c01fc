#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
run() {
  TT=$1; shift
  timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/e_$TT.json 2> $O/e_$TT.err || { tail -20 $O/e_$TT.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/e_$TT.json"))
k=d["kernel_ms"]
print("%-22s value=%.4g ms/step=%.4f" % ("$TT", d["value"], d["ms_per_step"]), {a: round(b,4) for a,b in k.items()})
PY
}
run cfg2 --workload cfg2
run cfg3_pipe2 --pipeline 2
run cfg3 
