#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_pass or small_sets or edge or synthetic_vs_oracle" > $O/pytest_j.log 2>&1 || { tail -60 $O/pytest_j.log; exit 1; }
tail -1 $O/pytest_j.log
SELHIP_LIB=scripts/microbench/libselhip_trace.so timeout -k 10 200 python scripts/small_trace.py > $O/small_trace_j.txt 2>&1 && cat $O/small_trace_j.txt
run() {
  TT=$1; shift
  timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/j_$TT.json 2> $O/j_$TT.err || { tail -20 $O/j_$TT.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/j_$TT.json"))
k=d["kernel_ms"]
print("%-22s value=%.4g ms/step=%.4f" % ("$TT", d["value"], d["ms_per_step"]), {a: round(b,4) for a,b in k.items()}, d["roofline"].get("bound"))
PY
}
run cfg2_small --workload cfg2
run cfg2_regular --workload cfg2 --param small_pass=0
