#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -m gpu > $O/pytest_g.log 2>&1 || { tail -40 $O/pytest_g.log; exit 1; }
tail -1 $O/pytest_g.log
run() {
  TT=$1; shift
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/g_$TT.json 2> $O/g_$TT.err || { tail -20 $O/g_$TT.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/g_$TT.json"))
k=d["kernel_ms"]
print("%-22s value=%.4g ms/step=%.4f" % ("$TT", d["value"], d["ms_per_step"]), {a: round(b,4) for a,b in k.items() if a in ("join","verify","group","hist","select")})
PY
}
for V in 0 1; do
  run pf${V}_cfg3 --param hist_prefetch=$V
  run pf${V}_hard --param hist_prefetch=$V --hard
  run pf${V}_cfg4 --param hist_prefetch=$V --workload cfg4 --steps 10 --warmup 2 --pipeline 0
  run pf${V}_cfg5 --param hist_prefetch=$V --workload cfg5 --steps 5 --warmup 2 --pipeline 0
done
run pf1_cfg3_b1024 --param hist_bs_blocks=1024
run pf1_cfg3_b4096 --param hist_bs_blocks=4096
run pf1_cfg3_run1 --param hist_run=1
run pf1_cfg3_run4 --param hist_run=4
