"""Link-prediction head: inner-product decoder (HIP GEMM) and the retrieval loss.

Mirror of (paths relative to the reference root):
  DecoderType, LinkPredictionDecoder   python/gigl/src/common/models/layers/decoder.py:10-70
      inner_product: scores = torch.mm(q, c.T)   (:64-66)  -> gigl_linear(q, c) (same NT GEMM, exact fp32 MFMA)
      hadamard_MLP uses torch_geometric.nn.models.MLP (third-party, un-vendored) -> not implemented
  LinkPredictionGNN                    python/gigl/src/common/models/pyg/link_prediction.py:33-60
  RetrievalLoss                        python/gigl/src/common/models/layers/loss.py:177-360
      calculate_batch_retrieval_loss :209-277, _mask_by_query_ids :279-305, _mask_by_candidate_ids :307-331
The loss is small dense tensor algebra that the reference itself writes in torch; it is restated 1:1 (same
masks, `finfo.min` masking, CrossEntropyLoss(reduction="sum") against eye) and pinned by the reference's
known-answer tests (tests/test_link_prediction.py restates loss_test.py:61-166 and decoder_test.py:44-62).
"""
from __future__ import annotations

from enum import Enum
from typing import List, Optional

import torch
import torch.nn as nn


class DecoderType(Enum):
    hadamard_MLP = "hadamard_MLP"
    inner_product = "inner_product"

    @classmethod
    def get_all_criteria(cls) -> List[str]:
        return [m.name for m in cls]


class _InnerProductFn(torch.autograd.Function):
    """scores[Q, C] = q[Q, D] @ c[C, D]^T on the fp32 MFMA GEMM; backward = two more GEMMs"""

    @staticmethod
    def forward(ctx, q, c, eng):
        q, c = q.contiguous(), c.contiguous()
        m = torch.tensor([q.shape[0]], dtype=torch.int32, device=q.device)
        ctx.eng = eng
        ctx.save_for_backward(q, c)
        return eng.linear(q, c, None, m, int(q.shape[0]), 0)

    @staticmethod
    def backward(ctx, ds):
        q, c = ctx.saved_tensors
        eng = ctx.eng
        ds = ds.contiguous()
        mq = torch.tensor([q.shape[0]], dtype=torch.int32, device=q.device)
        mc = torch.tensor([c.shape[0]], dtype=torch.int32, device=q.device)
        dq = eng.linear(ds, c.t().contiguous(), None, mq, int(q.shape[0]), 0)               # [Q,C]@[C,D]
        dc = eng.linear(ds.t().contiguous(), q.t().contiguous(), None, mc, int(c.shape[0]), 0)  # [C,Q]@[Q,D]
        return dq, dc, None


class LinkPredictionDecoder(nn.Module):
    def __init__(self, decoder_type: DecoderType = DecoderType.inner_product,
                 decoder_channel_list: Optional[List[int]] = None, **mlp_kwargs):
        super().__init__()
        self.decoder_type = decoder_type  # AttributeError on a non-enum, like the reference (`.value` below)
        self.decoder_channel_list = decoder_channel_list
        if self.decoder_type.value == "hadamard_MLP" and not isinstance(decoder_channel_list, list):
            raise ValueError("The decoder channel list must be provided when using 'hadamard_MLP' decoder, however "
                             f"you provided {decoder_channel_list}")
        if isinstance(decoder_channel_list, list) and len(decoder_channel_list) <= 1:
            raise ValueError("The decoder channel list must have length at least 2, however you provided a list of "
                             f"length {len(decoder_channel_list)}")
        if isinstance(decoder_channel_list, list) and decoder_channel_list[-1] != 1:
            raise ValueError("The last element in decoder channel list must be equal to 1, however you provided "
                             f"{decoder_channel_list[-1]}")
        if self.decoder_type.value == "hadamard_MLP":
            raise NotImplementedError("hadamard_MLP needs torch_geometric.nn.models.MLP (third-party); only the "
                                      "inner-product decoder is on this path")
        self.engine = None

    def forward(self, query_embeddings: torch.Tensor, candidate_embeddings: torch.Tensor) -> torch.Tensor:
        if self.engine is None:
            raise RuntimeError("LinkPredictionDecoder needs a HipEngine (decoder.engine = eng); no CPU fallback")
        return _InnerProductFn.apply(query_embeddings, candidate_embeddings, self.engine)


class LinkPredictionGNN(nn.Module):
    """encoder + decoder pair (link_prediction.py:33-60): forward = encoder embeddings, decode = scores"""

    def __init__(self, encoder: nn.Module, decoder: LinkPredictionDecoder):
        super().__init__()
        self._encoder = encoder
        self._decoder = decoder

    def forward(self, data, *args, **kwargs) -> torch.Tensor:
        return self._encoder(data, *args, **kwargs)

    def decode(self, query_embeddings: torch.Tensor, candidate_embeddings: torch.Tensor) -> torch.Tensor:
        return self._decoder(query_embeddings, candidate_embeddings)

    @property
    def encoder(self) -> nn.Module:
        return self._encoder

    @property
    def decoder(self) -> LinkPredictionDecoder:
        return self._decoder


class RetrievalLoss(nn.Module):
    def __init__(self, loss: Optional[nn.Module] = None, temperature: Optional[float] = None,
                 remove_accidental_hits: bool = False):
        super().__init__()
        self._loss = loss if loss is not None else nn.CrossEntropyLoss(reduction="sum")
        self._temperature = temperature
        if self._temperature is not None and self._temperature < 1e-12:
            raise ValueError("The temperature is expected to be greater than 1e-12, however you provided "
                             f"{self._temperature}")
        self._remove_accidental_hits = remove_accidental_hits

    def calculate_batch_retrieval_loss(self, scores: torch.Tensor,
                                       candidate_sampling_probability: Optional[torch.Tensor] = None,
                                       query_ids: Optional[torch.Tensor] = None,
                                       candidate_ids: Optional[torch.Tensor] = None,
                                       device: torch.device = torch.device("cpu")) -> torch.Tensor:
        num_queries, num_candidates = scores.shape[0], scores.shape[1]
        torch._assert(num_queries <= num_candidates,
                      "Number of queries should be less than or equal to number of candidates in a batch")
        labels = torch.eye(num_queries, num_candidates).to(device=device)
        duplicates = torch.zeros_like(labels).to(device=device)
        if self._temperature is not None:
            scores = scores / self._temperature
        if candidate_sampling_probability is not None:
            scores = scores - torch.log(torch.clamp(candidate_sampling_probability, min=1e-10)).type(scores.dtype)
        if query_ids is not None:
            duplicates = torch.maximum(duplicates, self._mask_by_query_ids(query_ids, num_queries, num_candidates,
                                                                            labels.dtype, device))
        if self._remove_accidental_hits:
            if candidate_ids is None:
                raise ValueError("When accidental hit removal is enabled, candidate ids must be supplied.")
            duplicates = torch.maximum(duplicates, self._mask_by_candidate_ids(candidate_ids, num_queries,
                                                                                labels.dtype, device))
        if query_ids is not None or self._remove_accidental_hits:
            scores = scores + (duplicates - labels) * torch.finfo(scores.dtype).min
        return self._loss(scores, target=labels)

    def _mask_by_query_ids(self, query_ids: torch.Tensor, num_queries: int, num_candidates: int, dtype: torch.dtype,
                           device: torch.device = torch.device("cpu")) -> torch.Tensor:
        query_ids = torch.unsqueeze(query_ids, 1)
        duplicates = torch.eq(query_ids, query_ids.T).type(dtype)
        if num_queries < num_candidates:
            padding_zeros = torch.zeros((num_queries, num_candidates - num_queries), dtype=dtype).to(device=device)
            return torch.cat((duplicates, padding_zeros), dim=1)
        return duplicates

    def _mask_by_candidate_ids(self, candidate_ids: torch.Tensor, num_queries: int, dtype: torch.dtype,
                               device: torch.device = torch.device("cpu")) -> torch.Tensor:
        positive_indices = torch.arange(num_queries).to(device=device)
        positive_candidate_ids = torch.gather(candidate_ids, 0, positive_indices).unsqueeze(1)
        all_candidate_ids = torch.unsqueeze(candidate_ids, 1)
        return torch.eq(positive_candidate_ids, all_candidate_ids.T).type(dtype)
