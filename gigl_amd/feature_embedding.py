"""Feature embedding layer: selected (categorical) feature columns go through an embedding table, the rest pass through.

Mirror of python/gigl/src/common/models/pyg/nn/models/feature_embedding.py:15-175 (BasicHomogeneousGNN applies it to the
node features before the interaction layer and the convolutions, homogeneous.py:112-115).  The reference reads column
positions, vocabulary sizes and padding values from its TensorFlow-Transform FeatureSchema; here the same facts are
passed in directly (`feature_columns`: name -> (first column, width) in the stored feature row, in schema order;
`vocab_sizes`: name -> number of ids), so the layer has no TFT dependency.  Semantics kept: ids are the stored float
values cast to int (+1 when the out-of-vocabulary id is -1); the `width` embeddings of a feature are averaged over the
non-padding entries (an all-zero embedding row = padding); output = [untouched columns | one emb_dim block per embedded
feature, in `features_to_embed` order].  The lookups are row gathers on the device (torch's embedding kernel).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn


class FeatureEmbeddingLayer(nn.Module):
    def __init__(self, features_to_embed: Dict[str, int], feature_columns: Dict[str, Tuple[int, int]],
                 vocab_sizes: Dict[str, int], feature_dim: int, aggregation: str = "mean",
                 oov_idx: Optional[int] = None, padding_idx: Optional[int] = None,
                 feature_padding_idx_map: Optional[Dict[str, int]] = None):
        super().__init__()
        if aggregation != "mean":
            raise NotImplementedError(f"Aggregation method {aggregation} is not supported")
        assert padding_idx is None or padding_idx >= 0, "padding_idx for embedding layer has to be >= 0"
        assert oov_idx is None or oov_idx >= -1, "oov_idx has to be >= -1"
        self._features_to_embed = dict(features_to_embed)
        self._columns = dict(feature_columns)
        self._plus_one = oov_idx == -1  # ids start at -1: shift everything by one for the table
        self._keep = [c for name, (s, w) in self._columns.items() if name not in self._features_to_embed
                      for c in range(s, s + w)]
        self.feature_embedding_layers = nn.ModuleDict()
        self._out_dim = int(feature_dim)
        for name, emb_dim in self._features_to_embed.items():
            pad = (feature_padding_idx_map or {}).get(name, padding_idx)
            if self._plus_one and pad is not None:
                pad += 1
            self.feature_embedding_layers[name] = nn.Embedding(int(vocab_sizes[name]), int(emb_dim), padding_idx=pad)
            self._out_dim += int(emb_dim) - int(self._columns[name][1])

    @property
    def out_dim(self) -> int:
        """width of the rows this layer returns (the convolutions' in_dim)"""
        return self._out_dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        parts = [x[:, self._keep]]
        for name in self._features_to_embed:
            s, w = self._columns[name]
            ids = x[:, s:s + w].long()
            if self._plus_one:
                ids = ids + 1
            emb = self.feature_embedding_layers[name](ids)            # [n, w, emb_dim]
            live = (emb != 0).all(dim=2)                              # padding rows are all-zero embeddings
            count = live.sum(dim=1, keepdim=True).to(emb.dtype)
            parts.append(emb.sum(dim=1) / torch.where(count == 0, torch.full_like(count, 1e-8), count))
        return torch.cat(parts, dim=1)
