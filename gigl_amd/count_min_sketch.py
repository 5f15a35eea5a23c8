"""Count-min sketch of candidate ids, resident in HBM (the Retrieval task's candidate-sampling correction).

Mirror of python/gigl/src/common/models/layers/count_min_sketch.py:11-120 (same class / method names and meaning):
the reference keeps a numpy table on the host and walks a tensor's ids one at a time through hash((item, i)) % width;
here the depth x width int32 table is a device tensor, `add_torch_long_tensor` / `estimate_torch_long_tensor` are one
kernel launch each (gigl_cms_add / gigl_cms_estimate, csrc/loss.hip) over the whole id tensor, and the cell of
(id, row) is CPython's tuple hash restated on the device — the table equals the reference's for the same integer ids.
"""
from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np
import torch

from ._lib import check
from .engine import default_engine


class CountMinSketch:
    def __init__(self, width: int = 2000, depth: int = 10, device=None):
        self.__width, self.__depth = int(width), int(depth)
        self.__total = 0
        self.__device = torch.device(device) if device is not None else None
        self.__table = None  # allocated on first use, on the device of the first id tensor (or cuda:0)

    def __engine(self, like=None):
        if self.__device is None:
            self.__device = like.device if (like is not None and like.is_cuda) else torch.device("cuda", 0)
        eng = default_engine(self.__device)
        if self.__table is None:
            self.__table = torch.zeros((self.__depth, self.__width), dtype=torch.int32, device=eng.device)
        return eng

    def add_torch_long_tensor(self, tensor: torch.Tensor) -> None:
        """every id of the tensor counts once (duplicates included)"""
        eng = self.__engine(tensor)
        ids = tensor.reshape(-1).to(device=eng.device, dtype=torch.int64).contiguous()
        check(eng._lib.gigl_cms_add(eng._ctx, C.c_void_p(self.__table.data_ptr()), self.__width, self.__depth,
                                    C.c_void_p(ids.data_ptr()), int(ids.numel())), eng._ctx)
        self.__total += int(ids.numel())

    def estimate_torch_long_tensor(self, tensor: torch.Tensor) -> torch.Tensor:
        """int64 estimates, one per id, on the sketch's device"""
        eng = self.__engine(tensor)
        ids = tensor.reshape(-1).to(device=eng.device, dtype=torch.int64).contiguous()
        out = torch.empty(ids.numel(), dtype=torch.int64, device=eng.device)
        check(eng._lib.gigl_cms_estimate(eng._ctx, C.c_void_p(self.__table.data_ptr()), self.__width, self.__depth,
                                         C.c_void_p(ids.data_ptr()), int(ids.numel()), C.c_void_p(out.data_ptr())),
              eng._ctx)
        return out

    def add(self, item: Any, delta: int = 1) -> None:
        for _ in range(int(delta)):
            self.add_torch_long_tensor(torch.tensor([int(item)], dtype=torch.int64))

    def estimate(self, item: Any) -> int:
        return int(self.estimate_torch_long_tensor(torch.tensor([int(item)], dtype=torch.int64)).item())

    def total(self) -> int:
        return self.__total

    def get_table(self) -> np.ndarray:
        self.__engine()
        return self.__table.cpu().numpy()


def calculate_in_batch_candidate_sampling_probability(frequency_tensor: torch.Tensor, total_cnt: int,
                                                      batch_size: int) -> torch.Tensor:
    """Q(candidate in batch) ~= batch_size * frequency / total_cnt, capped at 1 (count_min_sketch.py:98-120)"""
    return (batch_size * frequency_tensor.float() / total_cnt).clamp(max=1.0)
