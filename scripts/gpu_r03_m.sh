#!/bin/bash
# dense survivor graphs walked by candidate-row slice per XCD: parity, then the hard set with the slicing off / on, then the usual sets
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bitplane or variants or synthetic_vs_oracle or pipeline or interleave" > $O/pytest_m.log 2>&1 || { tail -60 $O/pytest_m.log; exit 1; }
tail -1 $O/pytest_m.log
run() {
  TT=$1; shift
  timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/m_$TT.json 2> $O/m_$TT.err || { tail -20 $O/m_$TT.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/m_$TT.json"))
k=d["kernel_ms"]
print("%-14s value=%.4g ms/step=%.4f" % ("$TT", d["value"], d["ms_per_step"]), {a: round(b,4) for a,b in k.items()})
PY
}
run hard_off --hard --param hist_dense_degree=-1
run hard_on --hard
run hard_on0 --hard --param hist_dense_degree=0
run cfg3_on0 --param hist_dense_degree=0
run cfg3 
run cfg4 --workload cfg4
run cfg4_on0 --workload cfg4 --param hist_dense_degree=0
