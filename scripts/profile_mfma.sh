#!/bin/bash
# rocprofv3 evidence for the MFMA filter kernel (csrc/device_bf_mfma.h): kernel trace + stats, then counter passes (each its own run;
# counters never combined with sys/hip/hsa traces).  usage: scripts/profile_mfma.sh <tag> [exp_bf_mfma.py args]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $R/scripts/exp_bf_mfma.py ${*:-1000000 1536 1024 10}"
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_VALU_MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counters_available.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
# PROFILE_MFMA_SHORT=1: the kernel trace and the three MFMA / LDS passes only
if [ "${PROFILE_MFMA_SHORT:-0}" = 1 ]; then
for PASS in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT/pmc_$N -- $CMD > $OUT/pmc_$N.log 2>&1 || echo "pass $N failed" >> $OUT/failed_passes.txt
done
else
for PASS in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
            "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES" \
            "SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32" \
            "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" \
            "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT/pmc_$N -- $CMD > $OUT/pmc_$N.log 2>&1 || echo "pass $N failed" >> $OUT/failed_passes.txt
done
fi
{
  echo "# rocprofv3 summary ($TAG): $CMD"; echo; echo '```'; grep "mfma path\|canonical" $OUT/trace.log; echo '```'; echo
  echo "MFMA counters this rocprofv3 knows: $(tr '\n' ' ' < $OUT/mfma_counters_available.txt)"; echo
  echo "## kernel trace (--kernel-trace --stats)"; python $R/scripts/summarize_prof.py $OUT/trace --last 3 bf_mfma
  for d in $OUT/pmc_*/; do echo; echo "## PMC $(basename $d)"; python $R/scripts/summarize_prof.py $d --last 3 bf_mfma | grep -v "^| \|^|--"; done
  [ -f $OUT/failed_passes.txt ] && { echo; echo "## passes rocprofv3 refused"; cat $OUT/failed_passes.txt; }
} > $OUT/summary.md 2>&1
find $OUT -name "*.csv" -size +4000k -delete
cat $OUT/summary.md
