#!/bin/bash
# Instruction mix of the search kernel under bench.py (two SQ counter passes only; quick).
# usage: scripts/profile_insts.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/insts_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu --no-side-configs --hostile-rows 0 --steps 3 $*"
for PASS in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT/pmc_$N -- $BENCH > $OUT/pmc_$N.log 2>&1
  grep '^{' $OUT/pmc_$N.log | tail -1 > $OUT/bench_line.json
done
{
  echo "# instruction mix ($TAG): $BENCH"; echo; echo '```'; cat $OUT/bench_line.json; echo '```'
  for d in $OUT/pmc_*/; do echo; echo "## PMC $(basename $d)"; python $R/scripts/summarize_prof.py $d --last 3 hnsw_search; done
} > $OUT/summary.md 2>&1
find $OUT -name "*.csv" -size +4000k -delete
cat $OUT/summary.md
