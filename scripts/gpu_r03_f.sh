#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -m gpu > $O/pytest_f.log 2>&1 || { tail -40 $O/pytest_f.log; exit 1; }
tail -1 $O/pytest_f.log
run() {
  TT=$1; shift
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/f_$TT.json 2> $O/f_$TT.err || { tail -20 $O/f_$TT.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/f_$TT.json"))
k=d["kernel_ms"]
print("%-22s value=%.4g ms/step=%.4f" % ("$TT", d["value"], d["ms_per_step"]), {a: round(b,4) for a,b in k.items() if a in ("sigbuild","join","verify","group","hist","select")})
PY
}
for V in 0 1; do
  run jv${V}_cfg3 --param join_verify=$V
  run jv${V}_cfg2 --param join_verify=$V --workload cfg2
  run jv${V}_hard --param join_verify=$V --hard
  run jv${V}_cfg4 --param join_verify=$V --workload cfg4 --steps 10 --warmup 2
done
