#!/bin/bash
# Run a list of stages on the GPU box, each under its own `timeout`, each logging into gpurun_out/<dir>/<stage>.txt and one
# line into summary.txt BEFORE it starts and when it ends — a stage that hangs costs its limit and no more, and the log says
# which one it was.   usage (inside gpurun):  bash scripts/gpu_stages.sh <dir> <stage-file>
# stage-file lines:  <name> <limit-seconds> <command ...>      (# comments allowed; ENV=val prefixes allowed in the command)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
export HNSW_GPU_WATCHDOG_S=${HNSW_GPU_WATCHDOG_S:-60}
: > $O/summary.txt
while IFS= read -r line; do
  case "$line" in ''|'#'*) continue;; esac
  read -r name limit cmd <<< "$line"
  echo "START $name (limit $limit s): $cmd" >> $O/summary.txt
  t0=$SECONDS
  timeout -k 10 $limit bash -c "$cmd" > $O/$name.txt 2>&1
  rc=$?
  echo "END   $name rc=$rc $((SECONDS - t0)) s" | tee -a $O/summary.txt
  tail -6 $O/$name.txt
done < "$2"
cat $O/summary.txt
