#!/bin/bash
# A/B runs of experiment builds of libhnsw_gpu.so (pg_embedding_amd/build.py variant <tag> DEFINES...) on one box:
# usage: scripts/exp_variants.sh <out-name> "<exp_cfg args>" tag [tag ...]     ("base" = the product library)
# Every line carries a CRC of labels + distance bits + E_q/H_q, so variants can be checked for identical results.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
ARGS=$1; shift
mkdir -p $(dirname $OUT)
: > $OUT
for tag in "$@"; do
  if [ "$tag" = base ]; then unset PGEMB_GPU_LIB; else export PGEMB_GPU_LIB=$R/pg_embedding_amd/lib/variants/libhnsw_gpu_$tag.so; fi
  case $tag in stamps*) export HNSW_GPU_TEAM_COUNTERS=1;; *) unset HNSW_GPU_TEAM_COUNTERS;; esac
  echo "## variant $tag: exp_cfg.py $ARGS" >> $OUT
  timeout 300 python $R/scripts/exp_cfg.py $ARGS >> $OUT 2>&1 || echo "variant $tag failed ($?)" >> $OUT
done
cat $OUT | tail -40
