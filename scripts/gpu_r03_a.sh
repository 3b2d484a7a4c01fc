#!/bin/bash
# round 3, first GPU call: bit-plane histogram kernel -- instruction rates, parity, A/B bench
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
BITPLANE_ONLY=1 timeout -k 10 300 scripts/microbench/valu_rate > $O/bitplane_rate.txt 2>&1 || { tail -5 $O/bitplane_rate.txt; exit 1; }
grep "W=4\|W=8" $O/bitplane_rate.txt | cut -c1-120
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bitplane or building_blocks or synthetic_vs_oracle or harder or stage2_grouping or histogram_variants or overflow or drop_in or edge" > $O/pytest_a.log 2>&1 || { tail -40 $O/pytest_a.log; exit 1; }
tail -2 $O/pytest_a.log
for A in "hist_algo=0" "hist_algo=1"; do
  for H in "" "--hard"; do
    timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --param $A $H > $O/bench_${A}_${H#--}.json 2> $O/bench_${A}_${H#--}.err || { tail -20 $O/bench_${A}_${H#--}.err; exit 1; }
    python - <<PY
import json
d=json.load(open("$O/bench_${A}_${H#--}.json"))
print("[$A $H]", "value=%.4g"%d["value"], "ms/step=%.4f"%d["ms_per_step"], {k: round(v,4) for k,v in d["kernel_ms"].items()}, "sel", d["config"]["selected_pairs"])
PY
  done
done
