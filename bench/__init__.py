"""bench — the measuring code behind `python bench.py` (the CLI at the repo root), one module per workload family:
common (JSON line, peaks, PMC passes, generators, launcher), cli (flags + dispatch), products (the N = 1 headline),
sharded (hash-partitioned MAG240M: RCCL ranks / emulated world), train, entries (inferencer / sampler), gat_lp, typed,
cpu_baseline (the only module that touches oracle/)."""
from .common import build_workload, cora_c1, rmat_edges_gpu  # noqa: F401  (tests and scripts build the bench's workloads)
