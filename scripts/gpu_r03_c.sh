#!/bin/bash
# round 3: parity subset + cfg3 / hard bench of the current build (tag $1), extra bench args after it
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
T=$1; shift
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bitplane or building_blocks or synthetic_vs_oracle or harder or stage2_grouping or histogram_variants or overflow or drop_in or edge" > $O/pytest_$T.log 2>&1 || { tail -40 $O/pytest_$T.log; exit 1; }
tail -1 $O/pytest_$T.log
run() {
  TT=$1; shift
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/c_$TT.json 2> $O/c_$TT.err || { tail -20 $O/c_$TT.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/c_$TT.json"))
k=d["kernel_ms"]
print("%-28s value=%.4g ms/step=%.4f group=%.4f hist=%.4f select=%.4f verify=%.4f join=%.4f" % ("$TT", d["value"], d["ms_per_step"], k.get("group",0), k.get("hist",0), k.get("select",0), k.get("verify",0), k.get("join",0)))
PY
}
run ${T}_cfg3 "$@"
run ${T}_cfg3_lab1 --param group_label=1 "$@"
run ${T}_hard --hard "$@"
run ${T}_hard_run8 --hard --param hist_run=8 "$@"
if [ -n "$BIG" ]; then
  run ${T}_cfg4 --workload cfg4 --steps 10 --warmup 2 "$@"
  run ${T}_cfg5 --workload cfg5 --steps 5 --warmup 2 "$@"
fi
