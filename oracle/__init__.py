"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Nothing in ``gigl_amd/`` may import this package.  Allowed importers: ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg (as the checker / timed CPU
baseline, never as the product).  See oracle/gigl_oracle.c for the parity status.
"""
from .oracle import (  # noqa: F401
    INVALID,
    build_csc,
    collate_reference,
    ensure_built,
    hash_permutation,
    sample_khop,
    union_build,
    xxh64_int32,
)
