"""Trainable message passing on HIP kernels: autograd wrappers + the batch graph container.

Mirror of what the reference gets from PyG for this path:
  torch_geometric.data.Data{x, edge_index}  (built by PygGraphBuilder.build, python/gigl/src/common/
      graph_builder/pyg_graph_builder.py:20-69, cast at data_loaders/utils.py:59-146)  -> GraphData
  SAGEConv forward/backward (PyG 2.5.3 MessagePassing.propagate + scatter-mean + Linear)       -> sage_conv
The forward kernels are gigl_gather_mean + gigl_linear; the backward is gigl_linear (two more GEMMs on
transposed operands) + gigl_gather_mean_backward.  torch supplies memory, autograd bookkeeping, the loss
and the optimiser — exactly the parts the reference also takes from torch.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .engine import HipEngine
from .engine import dev_i32


@dataclass
class GraphData:
    """homogeneous batch graph: `x` [n, D] fp32, `edge_index` [2, E] int64 (row 0 = src, row 1 = dst) — the
    fields the reference's models read from a PyG Data — plus the CSR-by-destination the kernels use"""
    x: torch.Tensor
    edge_index: torch.Tensor
    rowptr: Optional[torch.Tensor] = None  # int32 [n+1]
    col: Optional[torch.Tensor] = None     # int32 [E], sources, rows ascending
    n_dev: Optional[torch.Tensor] = None   # int32 [1] = n (device-side row count for the kernels)
    edge_attr: Optional[torch.Tensor] = None      # [E, De] fp32 in edge_index order (PyG Data.edge_attr)
    edge_attr_csr: Optional[torch.Tensor] = None  # the same rows in `col` order
    # batches built in HBM (hbm.ResidentGraph.graph_data) also say where their rows came from: the nodes' global ids, the
    # engine that holds the resident feature table, and the number of nodes of level <= k (level-ordered union graph) —
    # what an encoder needs to read stored rows in place and to compute a layer only for the rows the next one reads
    node_ids: Optional[torch.Tensor] = None       # int32 [n] (uint32 global ids)
    table: Optional[object] = None                # engine.HipEngine with the feature table
    levels: Optional[list] = None                 # host ints, levels[k] = nodes of level <= k; levels[-1] = n
    x_fn: Optional[object] = None                 # () -> x, when x was left out (see features())

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0]) if self.x is not None else int(self.rowptr.numel()) - 1

    def features(self) -> torch.Tensor:
        """x — gathered now when the batch was built without it (x_fn: batches whose consumer reads the stored rows in
        place and asks for the dense matrix only when it falls back to a whole-graph forward)"""
        if self.x is None:
            self.x = self.x_fn()
        return self.x

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    def to(self, device) -> "GraphData":
        device = torch.device(device)
        if self.rowptr is not None and (self.x is None or (self.x.device == device and self.x.dtype == torch.float32)):
            return self  # (already resident with its CSR: batches built in HBM)
        x = self.x.to(device=device, dtype=torch.float32).contiguous()
        ei = self.edge_index.to(device)
        ea = None if self.edge_attr is None else self.edge_attr.to(device=device, dtype=torch.float32).contiguous()
        g = GraphData(x=x, edge_index=ei, edge_attr=ea)
        if device.type == "cuda":
            g._build_csr()
        return g

    def _build_csr(self) -> None:
        n, dev = self.num_nodes, self.x.device
        src, dst = self.edge_index[0], self.edge_index[1]
        order = torch.argsort(dst * max(n, 1) + src)  # by destination, sources ascending
        self.col = src[order].to(torch.int32).contiguous()
        if self.edge_attr is not None:
            self.edge_attr_csr = self.edge_attr[order].contiguous()
        deg = torch.bincount(dst, minlength=n)
        rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rp[1:] = torch.cumsum(deg, 0)
        self.rowptr = rp.to(torch.int32).contiguous()
        if self.col.numel() == 0:
            self.col = torch.zeros(1, dtype=torch.int32, device=dev)
        self.n_dev = torch.tensor([n], dtype=torch.int32, device=dev)


@dataclass
class RowsView:
    """the rows one SAGE layer computes over a device-resident union graph (models.HipBatch): row i < *n_dev is
    col[rowptr[i] : rowend[i]]; `rows` bounds the launch and sizes the output (static, so nothing is read back from the
    device); gather_ids translates a column entry into a row of the source (union.nodes -> the resident feature table)"""
    rowptr: torch.Tensor
    rowend: Optional[torch.Tensor]
    col: torch.Tensor
    n_dev: torch.Tensor
    rows: int
    gather_ids: Optional[torch.Tensor] = None


class _SageConvFn(torch.autograd.Function):
    """y = act([mean_{j->i} h_j | h_i] @ [W_l | W_r]^T + b) over the rows of `g`: a GraphData (every row of the batch
    graph, packed CSR) or a RowsView (a prefix of a union graph's rows; h None = the engine's resident feature table,
    which takes no gradient)"""

    @staticmethod
    def forward(ctx, h, w_l, b_l, w_r, eng: HipEngine, g, act: int, aggr: str = "mean"):
        view = isinstance(g, RowsView)
        d = int(w_l.shape[1])
        n = g.rows if view else int(h.shape[0])
        if h is not None:
            h = h.contiguous()
        # (rows beyond *n_dev are never written and never read: the weight gradient runs over the real rows only)
        out = torch.empty((n, 2 * d), dtype=torch.float32, device=w_l.device) if view else None
        a = eng.gather_mean(h, d, g.gather_ids if view else None, g.rowptr, g.rowend if view else None, g.col, g.n_dev,
                            n, out=out, aggr=aggr)
        wcat = torch.cat([w_l, w_r], dim=1).contiguous()
        y = eng.linear(a, wcat, b_l, g.n_dev, n, act,
                       out=torch.empty((n, int(w_l.shape[0])), dtype=torch.float32, device=w_l.device) if view else None)
        ctx.eng, ctx.g, ctx.act, ctx.d, ctx.aggr = eng, g, act, d, aggr
        ctx.has_bias = b_l is not None
        ctx.src_rows = None if h is None else int(h.shape[0])
        ctx.save_for_backward(a, wcat, y, *([h] if aggr == "max" else []))
        return y

    @staticmethod
    def backward(ctx, dy):
        a, wcat, y = ctx.saved_tensors[:3]
        eng, g, d = ctx.eng, ctx.g, ctx.d
        n = int(a.shape[0])
        dy = dy.contiguous()
        dev = dy.device
        # dW[N, 2d] = dy^T[N, n] a[n, 2d] and db = column sums of dy, over the layer's real rows only, the relu mask
        # applied on the way in: one library launch (gigl_linear_weight_grad)
        dw, db = eng.linear_weight_grad(dy, a, g.n_dev, relu_y=y if ctx.act == 1 else None, want_bias=ctx.has_bias)
        dh = None
        if ctx.needs_input_grad[0]:
            if ctx.act == 1:
                dy = dy * (y > 0).to(dy.dtype)
            # da[n, 2d] = dy[n, N] @ wcat[N, 2d]  ->  linear(a = dy, w = wcat^T)
            da = eng.linear(dy, wcat.t().contiguous(), None, g.n_dev, n, 0)
            if ctx.aggr in ("mean", "sum") and d % 4 == 0 and ctx.src_rows >= n:
                # every source row written once by a gather over the transposed rows: no zero-fill, no float atomics
                dh = torch.empty((ctx.src_rows, d), dtype=torch.float32, device=dev)
                n_src = torch.full((1,), ctx.src_rows, dtype=torch.int32, device=dev)
                eng.gather_mean_backward_transposed(da, d, g.rowptr, getattr(g, "rowend", None), g.col, g.n_dev, n, n_src, dh,
                                                    aggr=ctx.aggr)
            else:
                dh = torch.zeros((ctx.src_rows, d), dtype=torch.float32, device=dev)
                eng.gather_mean_backward(da, d, g.rowptr, getattr(g, "rowend", None), g.col, g.n_dev, n, dh, aggr=ctx.aggr,
                                         src=ctx.saved_tensors[3] if ctx.aggr == "max" else None)
        return dh, dw[:, :d].contiguous(), db, dw[:, d:].contiguous(), None, None, None, None


def sage_conv(h: Optional[torch.Tensor], w_l: torch.Tensor, b_l: Optional[torch.Tensor], w_r: torch.Tensor, eng: HipEngine,
              g: GraphData, act: bool, aggr: str = "mean") -> torch.Tensor:
    return _SageConvFn.apply(h, w_l, b_l, w_r, eng, g, 1 if act else 0, aggr)
