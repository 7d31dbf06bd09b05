#!/bin/bash
# rocprofv3 evidence for ANY command: kernel trace + stats, then SQ counter passes (each its own run; counters are
# never combined with sys/hip/hsa traces).  Summaries of the LAST 3 dispatches of the search kernel.
# usage: scripts/profile_cmd.sh <tag> <command...>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="$*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
for PASS in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_LDS_ATOMIC SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH" \
            "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT/pmc_$N -- $CMD > $OUT/pmc_$N.log 2>&1
done
{
  echo "# rocprofv3 summary ($TAG): $CMD"; echo; echo '```'; grep -v "amdgpu.ids" $OUT/trace.log | tail -12; echo '```'; echo
  echo "## kernel trace (--kernel-trace --stats)"; python $R/scripts/summarize_prof.py $OUT/trace --last 3
  for d in $OUT/pmc_*/; do echo; echo "## PMC $(basename $d)"; python $R/scripts/summarize_prof.py $d --last 3 hnsw_search; done
} > $OUT/summary.md 2>&1
find $OUT -name "*.csv" -size +4000k -delete
cat $OUT/summary.md
