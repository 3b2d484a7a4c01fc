#!/bin/bash
# full GPU suite, smoke, then the default bench and the cfg2 lines (one-launch pass on / off)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_k.log 2>&1 || { tail -60 $O/pytest_k.log; exit 1; }
tail -1 $O/pytest_k.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
run() {
  TT=$1; shift
  timeout -k 10 400 python bench.py "$@" > $O/k_$TT.json 2> $O/k_$TT.err || { tail -20 $O/k_$TT.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/k_$TT.json"))
k=d["kernel_ms"]
print("%-14s value=%.4g ms/step=%.4f" % ("$TT", d["value"], d["ms_per_step"]), {a: round(b,4) for a,b in k.items()}, d["roofline"].get("bound"), d["roofline"].get("frac"))
PY
}
run default
run cfg2 --workload cfg2 --steps 50 --warmup 5 --no-cpu-baseline --no-extras
run cfg2_regular --workload cfg2 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --param small_pass=0
