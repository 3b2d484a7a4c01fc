#!/bin/bash
# round 3: sweep of the bit-plane histogram kernel's launch / grouping knobs (cfg3 and the hard variant)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03; mkdir -p $O
run() {  # tag, args...
  T=$1; shift
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > $O/sw_$T.json 2> $O/sw_$T.err || { tail -20 $O/sw_$T.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/sw_$T.json"))
k=d["kernel_ms"]
print("%-28s ms/step=%.4f group=%.4f hist=%.4f select=%.4f verify=%.4f" % ("$T", d["ms_per_step"], k.get("group",0), k.get("hist",0), k.get("select",0), k.get("verify",0)))
PY
}
for H in "" "--hard"; do
  for L in 0 1; do
    for R in 1 2 4 8; do
      run "lab${L}_run${R}${H#--}" --param group_label=$L --param hist_run=$R $H
    done
  done
  for B in 512 1024 4096 8192; do
    run "blocks${B}${H#--}" --param hist_bs_blocks=$B $H
  done
done
