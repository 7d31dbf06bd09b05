"""pg_embedding_amd — MI355X-native HNSW neighbor-scoring path behind pg_embedding's C boundary.

Only the hot path is here: the distance functions of distfunc.c and the
searchBaseLayer / searchKnn loop of hnswalg.cpp as hand-written gfx950 kernels in
``csrc/``, the C-ABI in ``include/``, and this thin host-side mirror.  There is no CPU
implementation: every entry point fails loudly without the built HIP library and a device.
"""
from .index import (  # noqa: F401
    DIST_L2, DIST_COSINE, DIST_MANHATTAN, OPCLASS, LABEL_DELETED, NO_LABEL,
    DEFAULT_M, DEFAULT_EF_CONSTRUCTION, DEFAULT_EF_SEARCH,
    GpuIndex, SearchContext, SearchStream, make_meta, dist_batch, l2_distance, cosine_distance, manhattan_distance,
    merge_topk_torch, merge_packed_torch, LocalShardedIndex,
)
from ._lib import HnswMetadata, LibraryMissing, config_set, config_get, sync_env  # noqa: F401

__version__ = "0.1.0"
