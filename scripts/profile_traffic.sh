#!/bin/bash
# HBM traffic of ANY command's search launches: FETCH_SIZE and WRITE_SIZE in their own rocprofv3 passes (counters are never combined with
# sys/hip/hsa traces), last 3 dispatches of the search kernel.  usage: scripts/profile_traffic.sh <tag> <command...>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="$*"
for PASS in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT/pmc_$PASS -- $CMD > $OUT/pmc_$PASS.log 2>&1
done
{
  echo "# HBM traffic ($TAG): $CMD"; echo '```'; grep "^dim" $OUT/pmc_FETCH_SIZE.log | cut -c1-330; echo '```'
  for d in $OUT/pmc_*/; do echo; echo "## PMC $(basename $d)"; python $R/scripts/summarize_prof.py $d --last 3 hnsw_search; done
  echo; echo "(gfx950: HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 for wide coalesced reads, MI355X_MICROARCH.md)"
} > $OUT/summary.md 2>&1
find $OUT -name "*.csv" -size +4000k -delete
cat $OUT/summary.md
