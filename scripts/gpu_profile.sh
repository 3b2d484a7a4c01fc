#!/bin/bash
# GPU box (via gpurun): one bench.py configuration -> bench line, rocprofv3 kernel trace + stats, PMC passes (own runs, no tracing).
#   scripts/gpu_profile.sh TAG [bench.py arguments...]        outputs under gpurun_out/prof_TAG/
# (counters are collected WITHOUT --kernel-trace/--stats in the same run; the program itself follows `--`)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 600 python3 bench.py "$@" > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 $R/bench.py --steps 10 --warmup 2 "$@" --no-cpu-baseline --no-extras > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_$N -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 "$@" --no-cpu-baseline --no-extras > $O/pmc_$N.log 2>&1 || { tail -20 $O/pmc_$N.log; exit 1; }
done
cd $R
find $O -name "*.csv" | head -20
rm -f $O/trace/*/*agent_info* 2>/dev/null
echo profile $TAG done
