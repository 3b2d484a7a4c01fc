#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -m gpu > $O/pytest_h.log 2>&1 || { tail -40 $O/pytest_h.log; exit 1; }
tail -1 $O/pytest_h.log
run() {
  TT=$1; shift
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/h_$TT.json 2> $O/h_$TT.err || { tail -20 $O/h_$TT.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/h_$TT.json"))
k=d["kernel_ms"]
print("%-22s value=%.4g ms/step=%.4f" % ("$TT", d["value"], d["ms_per_step"]), {a: round(b,4) for a,b in k.items() if a in ("sigbuild","join","verify","group","hist","select")})
PY
}
run cfg3
run cfg2 --workload cfg2
run hard --hard
run cfg4 --workload cfg4 --steps 10 --warmup 2 --pipeline 0
