"""CPU oracle for the HNSW hot path — TEST INFRASTRUCTURE ONLY.

Nothing under ``pg_embedding_amd/`` may import this package.  Allowed importers:
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Two checkers live here (see oracle/Makefile):

* :class:`PortIndex` — the plain-C restatement ``oracle/hnsw_port.c`` in the
  canonical (device) summation order; always buildable (gcc only).
* :class:`RefIndex`  — the UNMODIFIED reference ``distfunc.c`` + ``hnswalg.cpp``
  compiled from ``/root/reference`` into ``oracle/_ref/libpgemb_ref.so`` behind the
  flat-memory host ``oracle/flat_host.c``.  Present only where it was built (this
  container) or shipped prebuilt (the GPU box).
"""
from .bindings import (  # noqa: F401
    DIST_L2, DIST_COSINE, DIST_MANHATTAN,
    PortIndex, RefIndex, FlatHostIndex,
    build_oracle, have_ref, port_dist, port_dist_many, ref_dist, ref_dist_many,
    elem_size, lockstep_insert_compare,
)
