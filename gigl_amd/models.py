"""GNN encoders over the device-resident batch union graph.

Mirror of the reference's homogeneous PyG models for this path:
  GraphSAGE  python/gigl/src/common/models/pyg/homogeneous.py:157-202 (BasicHomogeneousGNN.forward :107-153)
Parameter names follow PyG 2.5.3's SAGEConv (`conv_layers.{i}.lin_l.{weight,bias}`,
`conv_layers.{i}.lin_r.weight`) so state_dicts interchange with the reference model
(names come from the un-vendored PyG: "parity unpinned").

Forward = layer-wise trimmed schedule on the level-ordered union graph (include/gigl_hip.h):
layer l of L only computes the rows that can still reach a root, always a prefix
[0, meta[LEVEL0 + L-1-l]).  For the roots this is exactly the reference's result, which runs every
layer over the whole union graph (SURVEY.md §0 fact 4, §7 "Batch-union semantics").
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import torch
import torch.nn as nn

from ._lib import GIGL_META_LEVEL0
from .engine import HipEngine, Tree, UnionGraph


@dataclass
class HipBatch:
    """what the HIP path hands to a model: the sampled trees and their union graph, all in HBM"""
    engine: HipEngine
    tree: Tree
    union: UnionGraph
    x: Optional[torch.Tensor] = None  # None: hydrate from the engine's resident feature table
    edge_attr: Optional[torch.Tensor] = None  # [cap_edges, De] rows aligned with union.col; None: engine.union_edge_attr
    x_index: Optional[torch.Tensor] = None  # int32 [cap]: row of `x` holding local node i (x is then NOT in node order)
    train: bool = False  # a training batch (gigl_amd/hbm.py): models run it with autograd when gradients are enabled

    @property
    def root_local(self) -> torch.Tensor:
        return self.union.root_local[: self.tree.b]

    def to(self, device=None, **_):
        """the batch already lives in HBM (the loops written for collated batches call batch.graph.to(device))"""
        return self


class SAGEConv(nn.Module):
    """parameter holder with PyG SAGEConv's layout: out = lin_l(mean_j x_j) + lin_r(x_i)"""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True, root_weight: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False) if root_weight else None

    def fused_weight(self) -> torch.Tensor:
        """[out, 2*in] = [W_l | W_r]: one projection of the [mean | self] operand"""
        wr = self.lin_r.weight if self.lin_r is not None else torch.zeros_like(self.lin_l.weight)
        return torch.cat([self.lin_l.weight, wr], dim=1).contiguous()


class JumpingKnowledge(nn.Module):
    """layer-wise representations -> one embedding: "cat" | "max" | "lstm" attention, then a Linear to out_dim
    (python/gigl/src/common/models/pyg/nn/models/jumping_knowledge.py:10-121; same parameter names: lstm, att,
    output_linear)"""

    def __init__(self, mode: str, hid_dim: int, out_dim: int, num_layers: Optional[int] = None,
                 lstm_dim: Optional[int] = None):
        super().__init__()
        self.mode = mode.lower()
        assert self.mode in ("cat", "max", "lstm")
        self.lstm = self.att = None
        if self.mode == "lstm":
            assert num_layers is not None, "num_layers cannot be None for lstm mode"
            lstm_dim = lstm_dim if lstm_dim else hid_dim
            self.lstm = nn.LSTM(input_size=hid_dim, hidden_size=(num_layers * lstm_dim) // 2, bidirectional=True,
                                batch_first=True)
            self.att = nn.Linear(2 * ((num_layers * lstm_dim) // 2), 1)
            self.output_linear = nn.Linear(hid_dim, out_dim)
        elif self.mode == "cat":
            assert num_layers is not None, "num_layers cannot be none for cat mode"
            self.output_linear = nn.Linear(num_layers * hid_dim, out_dim)
        else:
            self.output_linear = nn.Linear(hid_dim, out_dim)

    def forward(self, xs) -> torch.Tensor:
        if self.mode == "cat":
            return self.output_linear(torch.cat(xs, dim=-1))
        if self.mode == "max":
            return self.output_linear(torch.stack(xs, dim=-1).max(dim=-1)[0])
        x = torch.stack(xs, dim=1)  # [num_nodes, num_layers, hid_dim]
        alpha, _ = self.lstm(x)
        alpha = torch.softmax(self.att(alpha).squeeze(-1), dim=-1)
        return self.output_linear((x * alpha.unsqueeze(-1)).sum(dim=1))


class GraphSAGE(nn.Module):
    """BasicHomogeneousGNN + GraphSAGE.init_conv_layers (python/gigl/src/common/models/pyg/homogeneous.py:30-153,
    171-202): per layer conv -> [activation | BatchNorm1d | activation] -> dropout (not after the last layer unless
    activation_after_last_conv), optional L2 normalisation, return_emb, final Linear.  conv_kwargs: aggr ("mean" |
    "sum" | "max"), bias, root_weight (PyG SAGEConv); jk_mode ("cat" | "max" | "lstm") adds the JumpingKnowledge
    head over all layers' outputs (every conv then has hid_dim outputs and is followed by norm / activation).
    feature_embedding_layer (feature_embedding.FeatureEmbeddingLayer) and feature_interaction_layer (e.g.
    models_more.DCNv2) run on the node features before the first conv.
    State-dict keys follow the reference (conv_layers.{i}.lin_l/lin_r, batchnorm_layers.{i}, jk_layer.*, linear)."""

    def __init__(self, in_dim: int, hid_dim: int, out_dim: int, num_layers: int = 2,
                 activation_after_last_conv: bool = False, should_l2_normalize_embedding_layer_output: bool = False,
                 activation_before_norm: bool = False, dropout: float = 0.0, batchnorm: bool = False,
                 linear_layer: bool = False, return_emb: bool = False, jk_mode: Optional[str] = None,
                 jk_lstm_dim: Optional[int] = None, feature_interaction_layer: Optional[nn.Module] = None,
                 feature_embedding_layer: Optional[nn.Module] = None, **conv_kwargs):
        super().__init__()
        # selected columns through embedding tables (homogeneous.py:112-115; gigl_amd.feature_embedding), then the node
        # feature interaction (:117-119, e.g. models_more.DCNv2), both before the convolutions
        self.feature_embedding_layer = feature_embedding_layer
        self.feats_interaction = feature_interaction_layer
        conv_kwargs = dict(conv_kwargs.get("conv_kwargs") or conv_kwargs)
        self.in_dim, self.hid_dim, self.out_dim, self.num_layers = in_dim, hid_dim, out_dim, num_layers
        self.activation_after_last_conv = activation_after_last_conv
        self.activation_before_norm = activation_before_norm
        self.should_l2_normalize_embedding_layer_output = should_l2_normalize_embedding_layer_output
        self.aggr = str(conv_kwargs.get("aggr", "mean"))
        if self.aggr == "add":
            self.aggr = "sum"
        if self.aggr not in ("mean", "sum", "max"):
            raise NotImplementedError(f"SAGEConv aggr={self.aggr!r} is not implemented (mean, sum, max)")
        bias = bool(conv_kwargs.get("bias", True))
        root_weight = bool(conv_kwargs.get("root_weight", True))
        last = hid_dim if (linear_layer or jk_mode) else out_dim
        self.conv_layers = nn.ModuleList([
            SAGEConv(in_dim if i == 0 else hid_dim, hid_dim if i < num_layers - 1 else last, bias=bias,
                     root_weight=root_weight) for i in range(num_layers)])
        self.dropout = nn.Dropout(p=dropout)
        self.batchnorm = batchnorm
        if batchnorm:
            self.batchnorm_layers = nn.ModuleList([nn.BatchNorm1d(hid_dim)
                                                   for _ in range(num_layers if jk_mode else num_layers - 1)])
        self.jk_layer = (JumpingKnowledge(jk_mode, hid_dim, out_dim if not linear_layer else hid_dim, num_layers,
                                          jk_lstm_dim) if jk_mode else None)
        self.return_emb, self.linear_layer = return_emb, linear_layer
        if linear_layer:
            self.linear = nn.Linear(hid_dim, out_dim)
        self._ws = None  # workspace cache

    @property
    def _plain(self) -> bool:
        """conv -> relu only: what the fused kernels' epilogue and the one-call plan compute"""
        return (not self.batchnorm and self.dropout.p == 0.0 and not self.linear_layer
                and not self.activation_before_norm and self.jk_layer is None)

    def _post(self, h: torch.Tensor, l: int, fused_act: bool) -> torch.Tensor:
        """what follows conv l (homogeneous.py:126-141); fused_act: relu already applied by the kernel epilogue"""
        if l == self.num_layers - 1 and not self.activation_after_last_conv and self.jk_layer is None:
            return h
        if self._plain:
            return h if fused_act else torch.relu(h)
        if self.activation_before_norm:
            h = torch.relu(h)
        if self.batchnorm and l < len(self.batchnorm_layers):
            h = self.batchnorm_layers[l](h)
        if not self.activation_before_norm:
            h = torch.relu(h)
        return self.dropout(h)

    def _head(self, h: torch.Tensor) -> torch.Tensor:
        if self.should_l2_normalize_embedding_layer_output:
            h = torch.nn.functional.normalize(h, p=2, dim=1)
        if self.return_emb:
            return h
        return self.linear(h) if self.linear_layer else h

    def forward(self, batch, engine: Optional[HipEngine] = None) -> torch.Tensor:
        """HipBatch  -> inference over the level-ordered union graph (trimmed schedule, no autograd): returns
                        [cap, out_dim]; rows [0, n_level0) are the distinct roots' outputs (index with
                        batch.root_local for per-root rows in batch order)
        GraphData -> every layer over the whole batch graph with autograd (the reference's execution order,
                        homogeneous.py:107-153): returns [n, out_dim]"""
        from .nn import GraphData, sage_conv
        if isinstance(batch, GraphData):
            eng = engine or getattr(self, "engine", None)
            if eng is None:
                raise RuntimeError("GraphSAGE.forward(GraphData) needs the HipEngine (model.engine = eng)")
            h = self._interact(batch.x, eng)
            xs = []
            for l, conv in enumerate(self.conv_layers):
                fused = self._plain and (l < self.num_layers - 1 or self.activation_after_last_conv)
                w_r = conv.lin_r.weight if conv.lin_r is not None else torch.zeros_like(conv.lin_l.weight)
                h = sage_conv(h, conv.lin_l.weight, conv.lin_l.bias, w_r, eng, batch, fused, self.aggr)
                h = self._post(h, l, fused)
                xs.append(h)
            if self.jk_layer is not None:
                h = self.jk_layer(xs)
            return self._head(h)
        if batch.train and torch.is_grad_enabled():
            return self._forward_union_autograd(batch)
        with torch.no_grad():
            return self._forward_union(batch)

    def _forward_union_autograd(self, batch: HipBatch) -> torch.Tensor:
        """training over a batch sampled in HBM: the trimmed schedule of _forward_union with autograd — layer l
        computes the rows of level <= L-1-l (a prefix of the level-ordered union graph, at most b*(1 + f0 + ...) rows:
        the static bound sizes every buffer, the exact count stays on the device), the first layer reads the resident
        feature table through union.nodes.  Root rows and every parameter gradient equal the reference's
        every-layer-over-the-whole-union order (only root rows enter the loss)."""
        from .nn import RowsView, sage_conv
        eng, u, tree = batch.engine, batch.union, batch.tree
        L = self.num_layers
        assert u.hops == L, "one hop per layer"
        if batch.x is not None or self.feats_interaction is not None or self.feature_embedding_layer is not None or \
                self.batchnorm:
            raise NotImplementedError("training over HipBatch: resident feature table, no feature interaction layers, "
                                      "no batch norm (rows beyond the layer's count are padding)")
        h, xs = None, []
        for l, conv in enumerate(self.conv_layers):
            rows, width = 0, tree.b
            for i in range(L - l):
                rows += width
                width *= tree.fanouts[i] if i < L else 1
            rows = min(rows, int(u.nodes.numel()))
            n_rows = u.meta[GIGL_META_LEVEL0 + (L - 1 - l): GIGL_META_LEVEL0 + (L - l)]
            view = RowsView(u.rowptr, u.rowend, u.col, n_rows, rows, gather_ids=u.nodes if l == 0 else None)
            fused = self._plain and (l < L - 1 or self.activation_after_last_conv)
            w_r = conv.lin_r.weight if conv.lin_r is not None else torch.zeros_like(conv.lin_l.weight)
            h = sage_conv(h, conv.lin_l.weight, conv.lin_l.bias, w_r, eng, view, fused, self.aggr)
            h = self._post(h, l, fused)
            xs.append(h)
        if self.jk_layer is not None:
            raise NotImplementedError("training over HipBatch with JumpingKnowledge: use the GraphData route")
        return self._head(h)

    def _forward_union(self, batch: HipBatch) -> torch.Tensor:
        eng, u = batch.engine, batch.union
        L = self.num_layers
        assert u.hops == L, "one hop per layer"
        cap = int(u.nodes.numel())
        h = None
        xs = []
        batch = self._interacted(batch)
        for l, conv in enumerate(self.conv_layers):
            n_rows = u.meta[GIGL_META_LEVEL0 + (L - 1 - l): GIGL_META_LEVEL0 + (L - l)]
            d = conv.in_channels
            abuf = self._buf("a", l, cap, 2 * d)
            if l == 0 and batch.x is None:
                a = eng.gather_mean(None, d, u.nodes, u.rowptr, u.rowend, u.col, n_rows, cap, out=abuf, aggr=self.aggr)
            elif l == 0:
                a = eng.gather_mean(batch.x, d, batch.x_index, u.rowptr, u.rowend, u.col, n_rows, cap, out=abuf,
                                    aggr=self.aggr)
            else:
                a = eng.gather_mean(h, d, None, u.rowptr, u.rowend, u.col, n_rows, cap, out=abuf, aggr=self.aggr)
            fused = self._plain and (l < L - 1 or self.activation_after_last_conv)
            h = eng.linear(a, conv.fused_weight(), conv.lin_l.bias, n_rows, cap, 1 if fused else 0,
                           out=self._buf("h", l, cap, conv.out_channels))
            if not self._plain:
                h = self._post(h, l, fused).contiguous()
            xs.append(h)
        if self.jk_layer is not None:
            # layer l's output exists for the nodes of level <= L-1-l; the roots (a prefix of every layer's rows)
            # are the only nodes whose representation is complete at all layers: JK is taken over them
            n_roots = int(u.meta[GIGL_META_LEVEL0].item())
            out = torch.zeros((cap, self.jk_layer.output_linear.out_features), dtype=torch.float32, device=h.device)
            out[:n_roots] = self.jk_layer([x[:n_roots] for x in xs])
            h = out
        return self._head(h)

    def _interact(self, x: torch.Tensor, eng: HipEngine) -> torch.Tensor:
        if self.feature_embedding_layer is not None:
            x = self.feature_embedding_layer(x)
        if self.feats_interaction is None:
            return x
        self.feats_interaction.engine = eng  # (its products run on this model's engine and stream)
        return self.feats_interaction(x)

    def _interacted(self, batch: HipBatch) -> HipBatch:
        """the batch with the feature-interaction layer applied to its nodes' rows (a local fp32 matrix in node order)"""
        if self.feats_interaction is None and self.feature_embedding_layer is None:
            return batch
        eng, u = batch.engine, batch.union
        cap = int(u.nodes.numel())
        if batch.x is None:
            x = eng.gather_rows(u.nodes, u.meta[0:1], cap)
        else:
            x = batch.x if batch.x_index is None else batch.x[batch.x_index.long()]
        valid = (torch.arange(cap, device=x.device) < u.meta[0:1].to(torch.int64))[:, None]
        x = torch.where(valid, x, torch.zeros_like(x))
        return HipBatch(eng, batch.tree, u, x=self._interact(x, eng).contiguous(), edge_attr=batch.edge_attr)

    def projected_input_pays(self, eng: HipEngine) -> bool:
        """True when a first layer over projected rows moves fewer bytes per aggregated edge than one over the stored
        rows (fp32 rows of the first layer's width against the table's rows) — MAG240M's 768 fp16 -> 256; not
        ogbn-products' 100 fp32 -> 256"""
        from ._lib import DTYPE_F32
        esz = 4 if eng.feat_dtype == DTYPE_F32 else 2
        # (a table of a few thousand rows is projected in microseconds of kernel time but milliseconds of fixed
        # overhead — workspace allocation, synchronisation — which a pass of a handful of batches cannot amortise)
        return (self._plain and self.aggr in ("mean", "sum") and self.feats_interaction is None
                and self.feature_embedding_layer is None and eng.n_nodes >= (1 << 16)
                and self.conv_layers[0].out_channels * 4 < self.conv_layers[0].in_channels * esz)

    def make_plan(self, eng: HipEngine, b: int, fanouts: Sequence[int], groups: int = 1):
        """one-call pipeline (sample -> union -> this model's forward -> one row per root) for batches of
        `b` roots on `eng`; weights are snapshotted — call plan.set_weights(*model.fused_params()) after updates.
        groups > 1: each call takes groups*b roots and processes them as `groups` independent batches of b."""
        assert len(fanouts) == self.num_layers, "one hop per layer"
        if not (self._plain and not self.should_l2_normalize_embedding_layer_output and self.feats_interaction is None
                and self.feature_embedding_layer is None):
            raise NotImplementedError("the one-call plan computes conv -> relu layers only; use forward(HipBatch)")
        w, bs = self.fused_params()
        return eng.make_sage_plan(w, bs, b, fanouts, act_last=self.activation_after_last_conv, groups=groups,
                                  aggr=self.aggr)

    def fused_params(self):
        return ([c.fused_weight().detach() for c in self.conv_layers],
                [None if c.lin_l.bias is None else c.lin_l.bias.detach() for c in self.conv_layers])

    def _buf(self, kind: str, layer: int, rows: int, cols: int) -> torch.Tensor:
        if self._ws is None:
            self._ws = {}
        key = (kind, layer)
        t = self._ws.get(key)
        dev = self.conv_layers[0].lin_l.weight.device
        if t is None or t.shape[0] < rows or t.shape[1] != cols or t.device != dev:
            t = torch.empty((rows, cols), dtype=torch.float32, device=dev)
            self._ws[key] = t
        return t[:rows]
