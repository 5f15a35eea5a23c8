"""Plugin protocol of the trainer / inferencer components (same names, arguments and behaviour as the reference).

  BaseModelOperationsProtocol  python/gigl/src/common/types/model.py:9-24
  BaseTrainer                  python/gigl/src/training/v1/lib/base_trainer.py:16-37
  BaseInferencer, InferBatchResults, no_grad_eval
                               python/gigl/src/inference/v1/lib/base_inferencer.py:23-57
  EvalMetric(sCollection)      python/gigl/src/common/types/model_eval_metrics.py:6-57
  hit_rate_at_k, mean_reciprocal_rank   python/gigl/src/common/utils/eval_metrics.py:6-73
"""
from __future__ import annotations

import importlib
from dataclasses import dataclass
from enum import Enum
from functools import wraps
from typing import Dict, List, Optional

import torch


class EvalMetricType(Enum):
    mrr = "mrr"
    loss = "loss"
    hits = "hits"
    acc = "acc"

    @classmethod
    def get_all_criteria(cls) -> List[str]:
        return [m.name for m in cls]


@dataclass
class EvalMetric:
    name: str
    value: float

    @classmethod
    def from_eval_metric_type(cls, eval_metric_type: EvalMetricType, value: float):
        return cls(name=eval_metric_type.name, value=value)

    def __post_init__(self):
        self.value = float(self.value)


class EvalMetricsCollection:
    def __init__(self, metrics: Optional[List[EvalMetric]] = None):
        self._metrics: Dict[str, EvalMetric] = dict()
        self.add_metrics(metrics or [])

    @property
    def metrics(self) -> Dict[str, EvalMetric]:
        return self._metrics

    def add_metric(self, model_metric: EvalMetric):
        self._metrics[model_metric.name] = model_metric

    def add_metrics(self, metrics: List[EvalMetric]):
        for m in metrics:
            self.add_metric(m)

    def __repr__(self):
        return f"{self.__class__.__name__}({', '.join(str(m) for m in self._metrics.values())})"


@dataclass
class InferBatchResults:
    embeddings: Optional[torch.Tensor]
    predictions: Optional[torch.Tensor]


class BaseModelOperationsProtocol:
    @property
    def model(self) -> torch.nn.Module:
        raise NotImplementedError

    @model.setter
    def model(self, model: torch.nn.Module) -> None:
        raise NotImplementedError

    def init_model(self, gbml_config_pb_wrapper, state_dict=None) -> torch.nn.Module:
        raise NotImplementedError


class BaseTrainer(BaseModelOperationsProtocol):
    def train(self, gbml_config_pb_wrapper, device: torch.device, profiler=None) -> None:
        raise NotImplementedError

    def eval(self, gbml_config_pb_wrapper, device: torch.device) -> EvalMetricsCollection:
        raise NotImplementedError

    def setup_for_training(self) -> None:
        raise NotImplementedError

    @property
    def supports_distributed_training(self) -> bool:
        raise NotImplementedError


class BaseInferencer(BaseModelOperationsProtocol):
    def infer_batch(self, batch, device: torch.device = torch.device("cpu")) -> InferBatchResults:
        raise NotImplementedError


def no_grad_eval(f):
    """eval mode + no_grad around infer_batch, restoring the training flag (base_inferencer.py:29-49)"""
    @wraps(f)
    def wrapper(self, *args, **kwargs):
        was_training = self.model.training
        self.model.eval()
        with torch.no_grad():
            ret = f(self, *args, **kwargs)
        self.model.train(mode=was_training)
        return ret
    return wrapper


def import_obj(path: str):
    """os_utils.import_obj: dotted path -> object (plugin loading, training_process.py:122-139)"""
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def _positive_ranks(pos_scores: torch.Tensor, neg_scores: torch.Tensor) -> torch.Tensor:
    """rank of every positive among {itself} + the negatives, 1 = best: one more than the number of negatives scoring
    strictly higher (a tie counts for the positive; the reference sorts [pos | negs] and leaves ties to argsort)"""
    return 1 + (neg_scores.reshape(1, -1) > pos_scores.reshape(-1, 1)).sum(dim=1)


def hit_rate_at_k(pos_scores: torch.Tensor, neg_scores: torch.Tensor, ks: torch.Tensor) -> torch.Tensor:
    """Hits@k for every k of `ks` (python/gigl/src/common/utils/eval_metrics.py:6-48; known answers restated in
    tests/test_reference_host_answers.py): the fraction of positives ranked within the top k of [positive | negatives];
    a k beyond 1 + #negatives is always a hit"""
    if int(torch.min(ks).item()) < 1:
        raise AssertionError(f"ks must be greater-or-equal to 1 (got {int(torch.min(ks).item())})")
    ranks = _positive_ranks(pos_scores, neg_scores)
    return (ranks.reshape(1, -1) <= ks.reshape(-1, 1).to(ranks.device)).float().mean(dim=1)


def mean_reciprocal_rank(pos_scores: torch.Tensor, neg_scores: torch.Tensor) -> torch.Tensor:
    """mean of 1 / rank over the positives (eval_metrics.py:51-73)"""
    return (1.0 / _positive_ranks(pos_scores, neg_scores).float()).mean()
