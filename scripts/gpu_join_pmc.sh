#!/bin/bash
# GPU box: PMC passes over join_probe.py for the variants given as "Q WPB QT" strings; prints per-kernel averages of the join
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=$1; shift
mkdir -p $R/gpurun_out/r02
export TMPDIR=/tmp
cd /tmp
for V in "$@"; do
  TAG=$(echo $V | tr ' ' '_')
  for SET in "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE32_INSTS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
    ST=$(echo $SET | cut -c1-12 | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/r02/pmc_${WL}_${TAG}_$ST -o pmc -- python3 $R/scripts/join_probe.py $WL $V > $R/gpurun_out/r02/pmc_${WL}_${TAG}_$ST.log 2>&1 || { tail -5 $R/gpurun_out/r02/pmc_${WL}_${TAG}_$ST.log; }
  done
done
cd $R
python3 - <<PY
import csv, collections, glob
for d in sorted(glob.glob("gpurun_out/r02/pmc_${WL}_*/")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if "join" in k: print(d.split("/")[-2], k[:30], {c: "%.4g"%(sum(x)/len(x)) for c,x in v.items()})
PY
