#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_pass or small_sets or edge or synthetic_vs_oracle" > $O/pytest_i.log 2>&1 || { tail -60 $O/pytest_i.log; exit 1; }
tail -1 $O/pytest_i.log
run() {
  TT=$1; shift
  timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/i_$TT.json 2> $O/i_$TT.err || { tail -20 $O/i_$TT.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/i_$TT.json"))
k=d["kernel_ms"]
print("%-22s value=%.4g ms/step=%.4f" % ("$TT", d["value"], d["ms_per_step"]), {a: round(b,4) for a,b in k.items()}, d["roofline"].get("bound"))
PY
}
run cfg2_small --workload cfg2
run cfg2_regular --workload cfg2 --param small_pass=0
run cfg2_cb --workload cfg2 --mode CB+smh_a
run cfg2_coop --workload cfg2 --param small_pass=2
