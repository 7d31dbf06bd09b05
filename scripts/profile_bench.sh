#!/bin/bash
# Collect the rocprofv3 evidence for bench.py: kernel trace + stats, then PMC passes
# (each in its own run; counters never combined with sys/hip/hsa traces).
# usage: scripts/profile_bench.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu --no-side-configs --hostile-rows 0 --serial-rows 0 --steps 3 $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_line.json
for PASS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT/pmc_$N -- $BENCH > $OUT/pmc_$N.log 2>&1
done
{
  echo "# rocprofv3 summary ($TAG): $BENCH"; echo; echo '```'; cat $OUT/bench_line.json; echo '```'; echo
  echo "## kernel trace (--kernel-trace --stats)"; python $R/scripts/summarize_prof.py $OUT/trace --last 3
  for d in $OUT/pmc_*/; do echo; echo "## PMC $(basename $d)"; python $R/scripts/summarize_prof.py $d --last 3 hnsw_search; done
} > $OUT/summary.md 2>&1
python $R/scripts/make_traffic_json.py $OUT > $OUT/traffic.json
# keep only the small files for merging back
find $OUT -name "*.csv" -size +4000k -delete
cat $OUT/summary.md; cat $OUT/traffic.json
