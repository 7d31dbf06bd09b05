#!/bin/bash
# rocprofv3 evidence PER CONFIGURATION of bench.py (M = the headline, C2, C3, C5, hostile): kernel trace + stats, FETCH_SIZE, WRITE_SIZE,
# one SQ pass and (best effort) the address-translation counters — each pass its own run of `bench.py --profile-config <cfg>`
# (counters are never combined with sys/hip/hsa traces), summaries of the LAST --steps dispatches of the search kernel, and
# profiles/traffic.json-style entries keyed by configuration.
# usage: scripts/profile_configs.sh <tag> [cfg ...]        (default: M C2 C3 C5 hostile)
set -u
TAG=$1; shift
CFGS=${*:-M C2 C3 C5 hostile}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cp $R/profiles/traffic.json $OUT/traffic.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
for CFG in $CFGS; do
  D=$OUT/$CFG; mkdir -p $D
  CMD="python $R/bench.py --profile-config $CFG --steps 3"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- $CMD > $D/trace.log 2>&1
  grep '^{"' $D/trace.log | tail -1 > $D/line.json
  for PASS in "FETCH_SIZE" "WRITE_SIZE" \
              "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
              "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
              "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do
    N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
    # (counters for the search kernels only: with every dispatch of an 8M-row build counted, rocprofv3 itself crashed — r6a, r6g)
    timeout 900 rocprofv3 --kernel-trace --pmc $PASS --kernel-include-regex "hnsw_search_kernel" --output-format csv -d $D/pmc_$N -- $CMD > $D/pmc_$N.log 2>&1
  done
  {
    echo "# rocprofv3 summary ($TAG, configuration $CFG): $CMD"; echo; echo '```'; cat $D/line.json; echo '```'; echo
    echo "## kernel trace (--kernel-trace --stats)"; python $R/scripts/summarize_prof.py $D/trace --last 3
    for d in $D/pmc_*/; do echo; echo "## PMC $(basename $d)"; python $R/scripts/summarize_prof.py $d --last 3 hnsw_search; done
  } > $OUT/${CFG}_summary.md 2>&1
  python $R/scripts/make_traffic_json.py $D $OUT/traffic.json > $D/traffic_entry.json 2>$D/traffic_entry.err
  find $D -name "*.csv" -size +2000k -delete
  tail -30 $OUT/${CFG}_summary.md
done
cat $OUT/traffic.json
