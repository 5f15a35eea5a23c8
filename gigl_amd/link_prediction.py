"""Link-prediction head: inner-product decoder (HIP GEMM) and the retrieval loss.

Mirror of (paths relative to the reference root):
  DecoderType, LinkPredictionDecoder   python/gigl/src/common/models/layers/decoder.py:10-70
      inner_product: scores = torch.mm(q, c.T)   (:64-66)  -> gigl_linear(q, c) (same NT GEMM, exact fp32 MFMA)
      hadamard_MLP: scores = MLP(q.unsqueeze(1) * c).sum(-1)  (:67-69, PyG MLP, un-vendored: "parity unpinned")
          -> the first layer of all pairs as one projection by query-scaled weights, the rest on gigl_linear
  LinkPredictionGNN                    python/gigl/src/common/models/pyg/link_prediction.py:33-60
  RetrievalLoss                        python/gigl/src/common/models/layers/loss.py:177-360
      calculate_batch_retrieval_loss :209-277, _mask_by_query_ids :279-305, _mask_by_candidate_ids :307-331
The loss is one fused device pass over the score matrix (csrc/loss.hip: temperature, sampling-probability correction,
both masks, log-softmax and the summed cross-entropy, plus its backward), pinned by the reference's known-answer tests
(tests/test_link_prediction.py restates loss_test.py:61-166 and decoder_test.py:44-62).
"""
from __future__ import annotations

from enum import Enum
from typing import List, Optional

import torch
import torch.nn as nn
from .engine import dev_i32


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
        m = dev_i32(q.device, q.shape[0])
        ctx.eng = eng
        ctx.save_for_backward(q, c)
        return eng.linear(q, c, None, m, int(q.shape[0]), 0)

    @staticmethod
    def backward(ctx, ds):
        q, c = ctx.saved_tensors
        eng = ctx.eng
        ds = ds.contiguous()
        mq = dev_i32(q.device, q.shape[0])
        mc = dev_i32(q.device, c.shape[0])
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
            self.mlp_decoder = _DecoderMLP(decoder_channel_list, **mlp_kwargs)
        self.engine = None

    def forward(self, query_embeddings: torch.Tensor, candidate_embeddings: torch.Tensor) -> torch.Tensor:
        if self.engine is None:
            raise RuntimeError("LinkPredictionDecoder needs a HipEngine (decoder.engine = eng); no CPU fallback")
        if self.decoder_type.value == "hadamard_MLP":
            return self.mlp_decoder.pair_scores(query_embeddings, candidate_embeddings, self.engine)
        return _InnerProductFn.apply(query_embeddings, candidate_embeddings, self.engine)


class _DecoderMLP(nn.Module):
    """the hadamard_MLP decoder's MLP with PyG MLP's parameter layout (`lins.{i}`; decoder.py:52-60 passes act,
    act_first, bias, plain_last, norm) evaluated on every (query, candidate) pair WITHOUT materialising the
    [Q*C, D] matrix of elementwise products: lin_0(q * c) = (W_0 * q) c, so the first layer of all pairs is ONE
    projection of the candidates by the [Q*h_1, D] matrix of query-scaled weights; the remaining layers run on the
    resulting [C*Q, h_1] rows.  score = sum over the last layer's (single) output."""

    def __init__(self, channel_list: List[int], act=torch.relu, act_first: bool = False, bias=False,
                 plain_last: bool = False, norm=None):
        super().__init__()
        if norm is not None:
            raise NotImplementedError("hadamard_MLP with a normalisation layer is not built")
        if isinstance(act, str):
            if act != "relu":
                raise NotImplementedError(f"hadamard_MLP activation {act!r} is not built (relu)")
            act = torch.relu
        n = len(channel_list) - 1
        biases = list(bias) if isinstance(bias, (list, tuple)) else [bool(bias)] * n
        if len(biases) != n:
            raise ValueError(f"Number of bias values provided ({len(biases)}) does not match the number of layers ({n})")
        self.lins = nn.ModuleList([nn.Linear(channel_list[i], channel_list[i + 1], bias=biases[i]) for i in range(n)])
        self.act, self.plain_last = act, plain_last

    def pair_scores(self, q: torch.Tensor, c: torch.Tensor, eng) -> torch.Tensor:
        from .models_hetero import _linear
        nq, nc = int(q.shape[0]), int(c.shape[0])
        n = len(self.lins)
        w0 = self.lins[0].weight                                     # [h1, D]
        h1 = int(w0.shape[0])
        scaled = (q.unsqueeze(1) * w0.unsqueeze(0)).reshape(nq * h1, -1)  # row (q, j) = W_0[j] * q
        x = _linear(eng, c, scaled, None).view(nc * nq, h1)            # row (c, q) = lin_0(q * c) before the bias
        if self.lins[0].bias is not None:
            x = x + self.lins[0].bias
        for i in range(n):
            if i > 0:
                x = _linear(eng, x, self.lins[i].weight, self.lins[i].bias)
            if self.act is not None and not (self.plain_last and i == n - 1):
                x = self.act(x)
        return x.view(nc, nq, -1).sum(dim=-1).t().contiguous()


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


def _ids_on(t: Optional[torch.Tensor], dev, n: int, what: str) -> Optional[torch.Tensor]:
    if t is None:
        return None
    t = t.to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
    if t.numel() != n:
        raise ValueError(f"{what}: expected {n} ids, got {t.numel()}")
    return t


class _FusedRetrievalLoss(torch.autograd.Function):
    """scores -> summed softmax cross-entropy against the diagonal, with the temperature, the sampling-probability
    correction and both masks applied inside the kernel (csrc/loss.hip); backward = one more pass over the scores"""

    @staticmethod
    def forward(ctx, scores, eng, temperature, cand_prob, query_ids, cand_ids):
        scores = scores.contiguous()
        loss, lse, _ = eng.retrieval_loss(scores, temperature, cand_prob, query_ids, cand_ids)
        ctx.eng, ctx.temperature = eng, temperature
        ctx.save_for_backward(scores, lse, cand_prob, query_ids, cand_ids)
        return loss

    @staticmethod
    def backward(ctx, g):
        scores, lse, cand_prob, query_ids, cand_ids = ctx.saved_tensors
        d = ctx.eng.retrieval_loss_backward(scores, ctx.temperature, cand_prob, query_ids, cand_ids, lse,
                                            g.to(torch.float32).contiguous())
        return d, None, None, None, None, None


class _MaskedLogits(torch.autograd.Function):
    """the masked logits themselves (for a caller-supplied loss module): the masks add constants, so the gradient is
    the incoming one over the temperature"""

    @staticmethod
    def forward(ctx, scores, eng, temperature, cand_prob, query_ids, cand_ids):
        _, _, masked = eng.retrieval_loss(scores.contiguous(), temperature, cand_prob, query_ids, cand_ids,
                                          want_masked=True)
        ctx.temperature = temperature
        return masked

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.temperature if ctx.temperature is not None else g), None, None, None, None, None


class RetrievalLoss(nn.Module):
    """Same constructor and `calculate_batch_retrieval_loss` signature as the reference's RetrievalLoss
    (python/gigl/src/common/models/layers/loss.py:177-277); the body is ONE fused device pass (gigl_retrieval_loss)
    instead of the mask tensors + CrossEntropyLoss.  A caller-supplied `loss` module receives the masked logits and
    eye(Q, C) targets, as in the reference.  Device tensors only: there is no CPU path."""

    def __init__(self, loss: Optional[nn.Module] = None, temperature: Optional[float] = None,
                 remove_accidental_hits: bool = False):
        super().__init__()
        if temperature is not None and temperature < 1e-12:
            raise ValueError("The temperature is expected to be greater than 1e-12, however you provided "
                             f"{temperature}")
        self._loss = loss  # None: summed cross-entropy, computed inside the kernel
        self._temperature = temperature
        self._remove_accidental_hits = remove_accidental_hits

    def calculate_batch_retrieval_loss(self, scores: torch.Tensor,
                                       candidate_sampling_probability: Optional[torch.Tensor] = None,
                                       query_ids: Optional[torch.Tensor] = None,
                                       candidate_ids: Optional[torch.Tensor] = None,
                                       device: torch.device = torch.device("cpu")) -> torch.Tensor:
        from .engine import default_engine
        if self._remove_accidental_hits and candidate_ids is None:
            raise ValueError("When accidental hit removal is enabled, candidate ids must be supplied.")
        if not scores.is_cuda:
            raise RuntimeError("RetrievalLoss runs on the HIP device only (scores is a CPU tensor); no CPU fallback")
        q, c = int(scores.shape[0]), int(scores.shape[1])
        if q > c:
            raise AssertionError("Number of queries should be less than or equal to number of candidates in a batch")
        dev = scores.device
        eng = default_engine(dev)
        scores = scores.to(torch.float32)
        prob = None
        if candidate_sampling_probability is not None:
            prob = candidate_sampling_probability.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
            if prob.numel() != c:  # (the reference's `scores - log(prob)` would fail to broadcast)
                raise ValueError(f"candidate_sampling_probability has {prob.numel()} entries for {c} candidates")
        qid = _ids_on(query_ids, dev, q, "query_ids")
        cid = _ids_on(candidate_ids, dev, c, "candidate_ids") if self._remove_accidental_hits else None
        if self._loss is None:
            return _FusedRetrievalLoss.apply(scores, eng, self._temperature, prob, qid, cid)
        masked = _MaskedLogits.apply(scores, eng, self._temperature, prob, qid, cid)
        return self._loss(masked, target=torch.eye(q, c, device=dev))
