"""Count-min sketch of the reference, restated (TEST INFRASTRUCTURE ONLY — imported by tests/, never by gigl_amd/).

Follows python/gigl/src/common/models/layers/count_min_sketch.py:11-95: a depth x width int32 table; item x touches
cell hash((x, i)) % width of row i (Python's built-in tuple hash — deterministic for ints — and Python's non-negative
remainder); add increments the depth cells, estimate takes their minimum.  Pinned by the reference's own known-answer
test (python/tests/unit/src/common/models/layers/count_min_sketch_test.py:12-24, restated in tests/test_cms.py).
"""
from __future__ import annotations

import numpy as np


class CountMinSketch:
    def __init__(self, width: int = 2000, depth: int = 10):
        self.width, self.depth = width, depth
        self.table = np.zeros((depth, width), dtype=np.int32)
        self.count = 0

    def add(self, item: int, delta: int = 1) -> None:
        for i in range(self.depth):
            self.table[i][hash((int(item), i)) % self.width] += delta
        self.count += delta

    def add_all(self, items) -> None:
        for x in np.asarray(items).reshape(-1).tolist():
            self.add(x)

    def estimate(self, item: int) -> int:
        return int(min(self.table[i][hash((int(item), i)) % self.width] for i in range(self.depth)))

    def estimate_all(self, items) -> np.ndarray:
        return np.array([self.estimate(x) for x in np.asarray(items).reshape(-1).tolist()], dtype=np.int64)

    def total(self) -> int:
        return self.count
