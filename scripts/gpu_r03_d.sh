#!/bin/bash
# round 3: join-form A/B (parity subset first)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "histogram_variants or synthetic_vs_oracle or overflow or harder or edge or interleaved or chunk_lanes" > $O/pytest_d.log 2>&1 || { tail -40 $O/pytest_d.log; exit 1; }
tail -1 $O/pytest_d.log
run() {
  TT=$1; shift
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/d_$TT.json 2> $O/d_$TT.err || { tail -20 $O/d_$TT.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/d_$TT.json"))
k=d["kernel_ms"]
print("%-22s value=%.4g ms/step=%.4f join=%.4f (timed %.4f) verify=%.4f hist=%.4f frac=%.3f" % ("$TT", d["value"], d["ms_per_step"], k.get("join",0), k.get("join_in_timed_region",0), k.get("verify",0), k.get("hist",0), (d["roofline"].get("frac") or 0)))
PY
}
for F in 0 1; do
  run tri${F}_cfg3 --param join_tri=$F
  run tri${F}_cfg3_qt64 --param join_tri=$F --param join_qt=64
  run tri${F}_cfg4 --param join_tri=$F --workload cfg4 --steps 10 --warmup 2 --pipeline 0
done
