#!/bin/bash
# short lists on one round of resident waves (stage 2a): parity subset, then the sizes it is meant for and the ones it must not touch
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bitplane or variants or synthetic_vs_oracle or small_sets or interleave or aux_hll" > $O/pytest_r.log 2>&1 || { tail -60 $O/pytest_r.log; exit 1; }
tail -1 $O/pytest_r.log
for W in 2 8; do timeout -k 10 200 python scripts/il_breakdown.py $W 2>&1 | grep -v amdgpu.ids; done
run() {
  TT=$1; shift
  timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/r_$TT.json 2> $O/r_$TT.err || { tail -20 $O/r_$TT.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/r_$TT.json"))
k=d["kernel_ms"]
print("%-14s value=%.4g ms/step=%.4f" % ("$TT", d["value"], d["ms_per_step"]), {a: round(b,4) for a,b in k.items()})
PY
}
run cfg3
run cfg2_regular --workload cfg2 --param small_pass=0
run n7000 --genomes 7000
run n5000 --genomes 5000
timeout -k 10 280 python scripts/emulate_strong.py cfg4 2 4 8 --sig-cache 2>&1 | grep -v amdgpu.ids > $O/strong_snake.txt; cat $O/strong_snake.txt
timeout -k 10 280 python scripts/emulate_strong.py cfg5 8 --sig-cache 2>&1 | grep -v amdgpu.ids > $O/strong5_snake.txt; cat $O/strong5_snake.txt
