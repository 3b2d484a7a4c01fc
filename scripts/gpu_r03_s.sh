#!/bin/bash
# PMC of stage 2a on one interleaved 1/8 share (28 280 genomes) and on the single-GPU workload
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/pmc_share; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for W in 8 1; do
  for SET in "FETCH_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    N=$(echo $SET | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/w${W}_$N -o pmc -- python3 $R/scripts/share_passes.py $W 0 > $O/w${W}_$N.log 2>&1 || { tail -20 $O/w${W}_$N.log; exit 1; }
  done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for W in (8, 1):
    for N in ("FETCH_SIZE", "SQ_INSTS_VALU"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"gpurun_out/r03/pmc_share/w{W}_{N}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "hll_union_hist_bs" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]]["v"].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            vals = v["v"]
            print("world", W, k, "launches", len(vals), "mean of last 3: %.4g" % (sum(vals[-3:]) / 3))
PY
