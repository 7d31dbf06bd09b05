#!/bin/bash
# The reference's unmodified glue (embedding.c on oracle/pgmock) over libembedding_gpu.so in process: the 1 650-row
# scenario with the validated mirror cache (default) and without it (PG_EMBEDDING_GPU_CACHE=0: a full walk + upload per call).
# usage: scripts/glue_cache_bench.sh   (on the GPU box; prints wall time and the cache counters)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
EXE=$(python -c "import sys; sys.path.insert(0, 'tests'); import server_util as SU; print(SU.build_pg_regress('gpu'))")
for name in scenario exhaust; do
  for cache in 1 0; do
    S=$(date +%s%N)
    PGEMB_PRINT_CACHE_STATS=1 PG_EMBEDDING_GPU_CACHE=$cache $EXE < tests/golden/pg_regress/$name.cmd > /tmp/glue_${name}_$cache.out 2> /tmp/glue_${name}_$cache.err
    E=$(date +%s%N)
    SAME=$(cmp -s /tmp/glue_${name}_$cache.out tests/golden/pg_regress/$name.expected && echo "same bytes as the reference" || echo "OUTPUT DIFFERS")
    echo "$name cache=$cache: $(( (E - S) / 1000000 )) ms wall, $SAME; $(grep 'shim cache' /tmp/glue_${name}_$cache.err)"
  done
done
