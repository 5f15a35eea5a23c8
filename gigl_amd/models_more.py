"""GIN and Transformer encoders of the reference's homogeneous model zoo on the path's kernels.

  GIN          python/gigl/src/common/models/pyg/homogeneous.py:205-249  (PyG GINConv over an MLP([in, o, o]))
  Transformer  python/gigl/src/common/models/pyg/homogeneous.py:440-487  (PyG TransformerConv, heads=1 on the last layer)
  GATv2        python/gigl/src/common/models/pyg/homogeneous.py:346-386  (PyG GATv2Conv, heads=1 on the last layer)
both under BasicHomogeneousGNN.forward (:107-153: activation / BatchNorm1d / dropout between layers, JumpingKnowledge,
L2 normalisation, return_emb, final Linear — shared with GraphSAGE here).  Parameter names follow PyG 2.5.3
(`conv_layers.{i}.nn.lins.{0,1}`, `conv_layers.{i}.nn.norms.0.module`, `conv_layers.{i}.eps`;
`conv_layers.{i}.lin_{query,key,value,skip,beta}`); PyG is not vendored by the reference: "parity unpinned".

No new kernels: a GIN layer is the segmented SUM + projection of a SAGE layer with the weights [W0 | (1+eps) W0]
(gigl_gather_reduce + gigl_linear, autograd through nn.sage_conv) followed by one more gigl_linear; a Transformer
layer is three projections + the dot-product attention reduce of HGTConv with one edge type (gigl_hgt_aggregate and
its backward).  GATv2 (homogeneous.py:346-386) has its own kernels (csrc/gatv2.hip: the logit is a C-wide pass per edge).
GINE (homogeneous.py:252-297) adds gigl_gine_aggregate (messages relu(x_j + lin(e_ji))).
GATv2(edge_dim) adds lin_edge(e) inside the logit's leaky_relu; Transformer(edge_dim) adds it to the keys and values
(gigl_transformer_aggregate_edge).

  DCNv2 / DCNCross  python/gigl/src/common/models/layers/feature_interaction.py:7-155 — the feature-interaction layer
BasicHomogeneousGNN applies to the node features before the first conv (`feature_interaction_layer=`): x_{i+1} =
x0 * (W x_i + b + diag_scale x_i) + x_i, W full or low-rank (U V); the products run on gigl_linear.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ._lib import GIGL_META_LEVEL0
from .engine import HipEngine, dev_i32
from .models import GraphSAGE, HipBatch


def dev_rows(t: torch.Tensor) -> torch.Tensor:
    """device-side row count of a matrix (the m_dev argument of gigl_linear)"""
    return dev_i32(t.device, int(t.shape[0]))


def _linear(eng: HipEngine, x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    from .models_hetero import _linear as lin
    return lin(eng, x, w, b)


class _PygBatchNorm(nn.Module):
    """torch_geometric.nn.norm.BatchNorm's parameter layout (`module` = BatchNorm1d)"""

    def __init__(self, channels: int):
        super().__init__()
        self.module = nn.BatchNorm1d(channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.module(x)


class _GinMLP(nn.Module):
    """PyG MLP([a, b, c], plain_last=True): lins.0 -> [act] -> norms.0 -> [act] -> lins.1"""

    def __init__(self, a: int, b: int, c: int, batchnorm: bool, act_first: bool):
        super().__init__()
        self.lins = nn.ModuleList([nn.Linear(a, b), nn.Linear(b, c)])
        self.norms = nn.ModuleList([_PygBatchNorm(b) if batchnorm else nn.Identity()])
        self.act_first, self.has_norm = act_first, batchnorm

    def tail(self, eng: HipEngine, h: torch.Tensor, relu_done: bool) -> torch.Tensor:
        """everything after lins.0 (relu_done: the projection's epilogue already applied the activation)"""
        if self.has_norm:
            if self.act_first:
                h = torch.relu(h)
            h = self.norms[0](h)
            if not self.act_first:
                h = torch.relu(h)
        elif not relu_done:
            h = torch.relu(h)
        return _linear(eng, h, self.lins[1].weight, self.lins[1].bias)


class GINConv(nn.Module):
    """out_i = nn((1 + eps) x_i + sum_{j->i} x_j)"""

    def __init__(self, in_channels: int, out_channels: int, eps: float = 0.0, train_eps: bool = False,
                 batchnorm: bool = False, act_first: bool = False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.nn = _GinMLP(in_channels, out_channels, out_channels, batchnorm, act_first)
        if train_eps:
            self.eps = nn.Parameter(torch.full((1,), float(eps)))
        else:
            self.register_buffer("eps", torch.full((1,), float(eps)))


class GIN(GraphSAGE):
    """conv_kwargs: eps, train_eps; batchnorm also puts a BatchNorm inside every conv's MLP (the reference passes
    norm="batch_norm" to PyG's MLP, homogeneous.py:236-239)"""

    def __init__(self, in_dim: int, hid_dim: int, out_dim: int, num_layers: int = 2, **kwargs):
        ck = dict(kwargs.get("conv_kwargs") or {})
        for key in ("eps", "train_eps"):
            if key in kwargs:
                ck[key] = kwargs.pop(key)
        kwargs.pop("conv_kwargs", None)
        super().__init__(in_dim, hid_dim, out_dim, num_layers=num_layers, **kwargs)
        last = hid_dim if (self.linear_layer or self.jk_layer is not None) else out_dim
        self.conv_layers = nn.ModuleList([
            GINConv(in_dim if i == 0 else hid_dim, hid_dim if i < num_layers - 1 else last, eps=float(ck.get("eps", 0.0)),
                    train_eps=bool(ck.get("train_eps", False)), batchnorm=self.batchnorm,
                    act_first=self.activation_before_norm) for i in range(num_layers)])
        self.aggr = "sum"

    def _conv_graph(self, conv: GINConv, h: torch.Tensor, eng: HipEngine, g) -> torch.Tensor:
        from .nn import sage_conv
        w0, b0 = conv.nn.lins[0].weight, conv.nn.lins[0].bias
        fused = not conv.nn.has_norm
        h = sage_conv(h, w0, b0, (1.0 + conv.eps) * w0, eng, g, fused, "sum")
        return conv.nn.tail(eng, h, fused)

    def forward(self, batch, engine: Optional[HipEngine] = None) -> torch.Tensor:
        from .nn import GraphData
        if isinstance(batch, GraphData):
            eng = engine or getattr(self, "engine", None)
            if eng is None:
                raise RuntimeError("GIN.forward(GraphData) needs the HipEngine (model.engine = eng)")
            h, xs = self._interact(batch.x, eng), []
            for l, conv in enumerate(self.conv_layers):
                h = self._post(self._conv_graph(conv, h, eng, batch), l, False)
                xs.append(h)
            if self.jk_layer is not None:
                h = self.jk_layer(xs)
            return self._head(h)
        with torch.no_grad():
            return self._forward_union(batch)

    def _forward_union(self, batch: HipBatch) -> torch.Tensor:
        eng, u = batch.engine, batch.union
        L = self.num_layers
        assert u.hops == L, "one hop per layer"
        cap = int(u.nodes.numel())
        h, xs = None, []
        for l, conv in enumerate(self.conv_layers):
            n_rows = u.meta[GIGL_META_LEVEL0 + (L - 1 - l): GIGL_META_LEVEL0 + (L - l)]
            d = conv.in_channels
            if l == 0:
                batch = self._interacted(batch)
            if l == 0 and batch.x is None:
                a = eng.gather_mean(None, d, u.nodes, u.rowptr, u.rowend, u.col, n_rows, cap, aggr="sum")
            elif l == 0:
                a = eng.gather_mean(batch.x, d, batch.x_index, u.rowptr, u.rowend, u.col, n_rows, cap, aggr="sum")
            else:
                a = eng.gather_mean(h, d, None, u.rowptr, u.rowend, u.col, n_rows, cap, aggr="sum")
            w0 = conv.nn.lins[0].weight
            fused = not conv.nn.has_norm
            h = eng.linear(a, torch.cat([w0, (1.0 + conv.eps) * w0], dim=1).contiguous(), conv.nn.lins[0].bias, n_rows,
                           cap, 1 if fused else 0)
            lin1 = conv.nn.lins[1]
            if conv.nn.has_norm:
                h = conv.nn.tail(eng, h, False)
            else:
                h = eng.linear(h, lin1.weight.contiguous(), lin1.bias, n_rows, cap, 0)
            h = self._post(h, l, False).contiguous()
            xs.append(h)
        if self.jk_layer is not None:
            n_roots = int(u.meta[GIGL_META_LEVEL0].item())
            out = torch.zeros((cap, self.jk_layer.output_linear.out_features), dtype=torch.float32, device=h.device)
            out[:n_roots] = self.jk_layer([x[:n_roots] for x in xs])
            h = out
        return self._head(h)

    def make_plan(self, *args, **kwargs):
        raise NotImplementedError("the one-call plan computes GraphSAGE layers only; use forward(HipBatch)")


class _GineAggFn(torch.autograd.Function):
    """gigl_gine_aggregate / gigl_gine_aggregate_backward over a CSR view: (1 + eps) x_i + sum_e relu(x_j + ee_e)"""

    @staticmethod
    def forward(ctx, x, ee, eps, eng, view, n_dev):
        x, ee, eps = x.contiguous(), ee.contiguous(), eps.to(torch.float32).contiguous()
        ctx.save_for_backward(x, ee, eps)
        ctx.meta = (eng, view, n_dev)
        return eng.gine_aggregate(x, ee, eps, view, n_dev)

    @staticmethod
    def backward(ctx, dout):
        x, ee, eps = ctx.saved_tensors
        eng, view, n_dev = ctx.meta
        dx, dee, deps = eng.gine_aggregate_backward(x, ee, eps, view, n_dev, dout)
        return dx, dee, (deps if ctx.needs_input_grad[2] else None), None, None, None


class GINEConv(GINConv):
    """out_i = nn((1 + eps) x_i + sum_{j->i} relu(x_j + lin(e_ji))); lin = Linear(edge_dim, in_channels) (PyG's name)"""

    def __init__(self, in_channels: int, out_channels: int, edge_dim: int, **kwargs):
        super().__init__(in_channels, out_channels, **kwargs)
        self.lin = nn.Linear(edge_dim, in_channels)


class GINE(GIN):
    """GIN with edge features in the messages (edge_dim required); conv_kwargs: eps, train_eps"""

    def __init__(self, in_dim: int, hid_dim: int, out_dim: int, num_layers: int = 2, edge_dim: Optional[int] = None,
                 **kwargs):
        if not edge_dim:
            raise ValueError("GINE needs edge_dim (the edge features' width)")
        super().__init__(in_dim, hid_dim, out_dim, num_layers=num_layers, **kwargs)
        self.edge_dim = int(edge_dim)
        old = list(self.conv_layers)
        self.conv_layers = nn.ModuleList([
            GINEConv(c.in_channels, c.out_channels, self.edge_dim, eps=float(c.eps.detach()), train_eps=isinstance(c.eps, nn.Parameter),
                     batchnorm=c.nn.has_norm, act_first=c.nn.act_first) for c in old])

    def _conv_rows(self, conv: GINEConv, h: torch.Tensor, edge_attr: torch.Tensor, eng: HipEngine, view, n_dev):
        ee = _linear(eng, edge_attr, conv.lin.weight, conv.lin.bias)
        agg = _GineAggFn.apply(h, ee, conv.eps, eng, view, n_dev)
        z = _linear(eng, agg, conv.nn.lins[0].weight, conv.nn.lins[0].bias)
        return conv.nn.tail(eng, z, False)

    def forward(self, batch, engine: Optional[HipEngine] = None) -> torch.Tensor:
        """GraphData (with edge_attr) -> every layer over the whole batch graph, autograd when grad mode is on
        HipBatch  -> the trimmed schedule over the union graph, edge rows from the resident edge-feature table"""
        from .models_attn import _CsrView
        from .nn import GraphData
        if isinstance(batch, GraphData):
            eng = engine or getattr(self, "engine", None)
            if eng is None:
                raise RuntimeError("GINE.forward(GraphData) needs the HipEngine (model.engine = eng)")
            if batch.edge_attr_csr is None:
                raise ValueError(f"the model was built with edge_dim={self.edge_dim} but the batch has no edge features")
            view, ea = _CsrView(batch), batch.edge_attr_csr
            if ea.shape[0] != view.col.numel():  # edgeless batch: col holds one padding entry
                ea = torch.zeros((view.col.numel(), self.edge_dim), dtype=torch.float32, device=batch.x.device)
            h, xs = self._interact(batch.x, eng), []
            for l, conv in enumerate(self.conv_layers):
                h = self._post(self._conv_rows(conv, h, ea, eng, view, batch.n_dev), l, False)
                xs.append(h)
            if self.jk_layer is not None:
                h = self.jk_layer(xs)
            return self._head(h)
        with torch.no_grad():
            eng, u = batch.engine, batch.union
            L = self.num_layers
            assert u.hops == L, "one hop per layer"
            cap = int(u.nodes.numel())
            ea = batch.edge_attr if batch.edge_attr is not None else eng.union_edge_attr(u)
            assert ea.shape[1] == self.edge_dim
            batch = self._interacted(batch)
            if batch.x is None:
                h = eng.gather_rows(u.nodes, u.meta[0:1], cap)
            else:
                h = batch.x if batch.x_index is None else batch.x[batch.x_index.long()].contiguous()
            valid = (torch.arange(cap, device=h.device) < u.meta[0:1].to(torch.int64))[:, None]
            h = torch.where(valid, h, torch.zeros_like(h))
            xs = []
            for l, conv in enumerate(self.conv_layers):
                n_dst = u.meta[GIGL_META_LEVEL0 + (L - 1 - l): GIGL_META_LEVEL0 + (L - l)]
                h = self._post(self._conv_rows(conv, h.contiguous(), ea, eng, u, n_dst), l, False).contiguous()
                xs.append(h)
            if self.jk_layer is not None:
                n_roots = int(u.meta[GIGL_META_LEVEL0].item())
                out = torch.zeros((cap, self.jk_layer.output_linear.out_features), dtype=torch.float32, device=h.device)
                out[:n_roots] = self.jk_layer([x[:n_roots] for x in xs])
                h = out
            return self._head(h)


class _TransformerEdgeAggFn(torch.autograd.Function):
    """gigl_transformer_aggregate_edge / its backward over a CSR view (rowptr / rowend / col)"""

    @staticmethod
    def forward(ctx, q, k, v, xe, eng, view, n_dev, heads, channels):
        q, k, v, xe = q.contiguous(), k.contiguous(), v.contiguous(), xe.contiguous()
        out = eng.transformer_aggregate_edge(q, k, v, xe, heads, channels, view, n_dev)
        ctx.save_for_backward(q, k, v, xe, out)
        ctx.meta = (eng, view, n_dev, heads, channels)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, xe, out = ctx.saved_tensors
        eng, view, n_dev, heads, channels = ctx.meta
        dq, dk, dv, dxe = eng.transformer_aggregate_edge_backward(q, k, v, xe, heads, channels, view, n_dev, out, dout)
        return dq, dk, dv, dxe, None, None, None, None, None


class TransformerConv(nn.Module):
    """parameter holder with PyG TransformerConv's layout (lin_key / lin_query / lin_value / lin_skip / lin_beta; with
    edge_dim also lin_edge, no bias: added to the keys and to the values of every edge)"""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True, beta: bool = False,
                 bias: bool = True, root_weight: bool = True, edge_dim: Optional[int] = None, dropout: float = 0.0):
        super().__init__()
        if dropout:
            raise NotImplementedError("TransformerConv attention dropout is not built")
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.root_weight, self.beta = concat, root_weight, beta and root_weight
        hc = heads * out_channels
        self.lin_key = nn.Linear(in_channels, hc)
        self.lin_query = nn.Linear(in_channels, hc)
        self.lin_value = nn.Linear(in_channels, hc)
        self.lin_edge = nn.Linear(edge_dim, hc, bias=False) if edge_dim is not None else None
        out_w = hc if concat else out_channels
        self.lin_skip = nn.Linear(in_channels, out_w, bias=bias) if root_weight else None
        self.lin_beta = nn.Linear(3 * out_w, 1, bias=False) if self.beta else None

    def forward(self, x: torch.Tensor, rowptr: torch.Tensor, col: torch.Tensor, eng: HipEngine,
                edge_attr: Optional[torch.Tensor] = None) -> torch.Tensor:
        from .models_hetero import _HgtAggFn
        n, H, C = int(x.shape[0]), self.heads, self.out_channels
        q = _linear(eng, x, self.lin_query.weight, self.lin_query.bias)
        k = _linear(eng, x, self.lin_key.weight, self.lin_key.bias)
        v = _linear(eng, x, self.lin_value.weight, self.lin_value.bias)
        if self.lin_edge is not None:
            from types import SimpleNamespace
            xe = _linear(eng, edge_attr, self.lin_edge.weight, None)
            view = SimpleNamespace(rowptr=rowptr, rowend=rowptr[1:], col=col)
            out = _TransformerEdgeAggFn.apply(q, k, v, xe, eng, view, dev_rows(x), H, C)
        else:
            p_rel = torch.ones((1, H), dtype=torch.float32, device=x.device)
            out = _HgtAggFn.apply(q, k, v, p_rel, eng, H, C, rowptr, col, None, n)
        if not self.concat:
            out = out.view(n, H, C).mean(1)
        if self.lin_skip is not None:
            xr = _linear(eng, x, self.lin_skip.weight, self.lin_skip.bias)
            if self.lin_beta is not None:
                b = torch.sigmoid(torch.cat([out, xr, out - xr], dim=-1) @ self.lin_beta.weight.t())
                out = b * xr + (1 - b) * out
            else:
                out = out + xr
        return out


class Transformer(GraphSAGE):
    """conv_kwargs: heads, beta, concat, bias, root_weight.  Head sizes the attention reduce is built for: channels a
    multiple of 4 with channels/4 a power of two <= 64, heads*channels <= 1024 (gigl_hgt_aggregate)."""

    def __init__(self, in_dim: int, hid_dim: int, out_dim: int, num_layers: int = 2, edge_dim: Optional[int] = None,
                 **kwargs):
        ck = dict(kwargs.get("conv_kwargs") or {})
        for key in ("heads", "beta", "concat", "bias", "root_weight", "dropout_attn"):
            if key in kwargs:
                ck[key] = kwargs.pop(key)
        kwargs.pop("conv_kwargs", None)
        heads = int(ck.get("heads", 1))
        super().__init__(in_dim, hid_dim, out_dim, num_layers=num_layers, **kwargs)
        last = hid_dim if (self.linear_layer or self.jk_layer is not None) else out_dim
        self.heads, self.edge_dim = heads, edge_dim
        self.conv_layers = nn.ModuleList([
            TransformerConv(in_dim if i == 0 else hid_dim * heads, hid_dim if i < num_layers - 1 else last,
                            heads=heads if i < num_layers - 1 else 1, concat=bool(ck.get("concat", True)),
                            beta=bool(ck.get("beta", False)), bias=bool(ck.get("bias", True)),
                            root_weight=bool(ck.get("root_weight", True)), edge_dim=edge_dim)
            for i in range(num_layers)])
        if self.batchnorm:  # BatchNorm1d(hid_dim * num_heads), homogeneous.py:78-87
            self.batchnorm_layers = nn.ModuleList([nn.BatchNorm1d(hid_dim * heads)
                                                   for _ in range(len(self.batchnorm_layers))])

    def _layers(self, x: torch.Tensor, rowptr: torch.Tensor, col: torch.Tensor, eng: HipEngine,
                edge_attr: Optional[torch.Tensor] = None) -> torch.Tensor:
        h, xs = x, []
        for l, conv in enumerate(self.conv_layers):
            h = self._post(conv(h, rowptr, col, eng, edge_attr), l, False)
            xs.append(h)
        if self.jk_layer is not None:
            h = self.jk_layer(xs)
        return self._head(h)

    def forward(self, batch, engine: Optional[HipEngine] = None) -> torch.Tensor:
        """GraphData -> every layer over the whole batch graph (autograd when grad mode is on): [n, out_dim]
        HipBatch  -> the same over the whole union graph (no trimmed schedule: the attention reduce takes a plain
                     CSR): [cap, out_dim], index with batch.root_local"""
        from .nn import GraphData
        if isinstance(batch, GraphData):
            eng = engine or getattr(self, "engine", None)
            if eng is None:
                raise RuntimeError("Transformer.forward(GraphData) needs the HipEngine (model.engine = eng)")
            ea = None
            if self.edge_dim is not None:
                if batch.edge_attr_csr is None:
                    raise ValueError(f"the model was built with edge_dim={self.edge_dim} but the batch has no edge features")
                ea = batch.edge_attr_csr
                if ea.shape[0] != batch.col.numel():  # edgeless batch: col holds one padding entry
                    ea = torch.zeros((batch.col.numel(), self.edge_dim), dtype=torch.float32, device=batch.x.device)
            return self._layers(self._interact(batch.x, eng), batch.rowptr, batch.col, eng, ea)
        with torch.no_grad():
            eng, u = batch.engine, batch.union
            cap = int(u.nodes.numel())
            n_nodes = u.meta[0:1]
            x = batch.x if batch.x is not None else eng.gather_rows(u.nodes, n_nodes, cap)
            if batch.x is not None and batch.x_index is not None:
                x = batch.x[batch.x_index.long()]
            # plain CSR over all `cap` rows: the union's rows are [rowptr[i], rowend[i]) slices of col
            deg = (u.rowend[:cap] - u.rowptr[:cap]).clamp(min=0).to(torch.int64)
            deg = deg * (torch.arange(cap, device=deg.device) < n_nodes.to(torch.int64)).to(torch.int64)
            rowptr = torch.zeros(cap + 1, dtype=torch.int64, device=deg.device)
            rowptr[1:] = torch.cumsum(deg, 0)
            e = int(rowptr[-1].item())
            pos = torch.arange(e, device=deg.device)
            row = torch.searchsorted(rowptr[1:], pos, right=True)
            at = u.rowptr.to(torch.int64)[row] + (pos - rowptr[row])  # positions in the union's col
            col = u.col[at].to(torch.int32).contiguous()
            ea = None
            if self.edge_dim is not None:
                ea_all = batch.edge_attr if batch.edge_attr is not None else eng.union_edge_attr(u)
                ea = ea_all[at].contiguous() if e else torch.zeros((1, self.edge_dim), device=deg.device)
            if e == 0:
                col = torch.zeros(1, dtype=torch.int32, device=deg.device)
            x = torch.where((torch.arange(cap, device=x.device) < n_nodes.to(torch.int64))[:, None], x, torch.zeros_like(x))
            x = self._interact(x, eng)
            return self._layers(x.contiguous(), rowptr.to(torch.int32).contiguous(), col, eng, ea)

    def make_plan(self, *args, **kwargs):
        raise NotImplementedError("the one-call plan computes GraphSAGE layers only; use forward(HipBatch)")


class _Gatv2AggFn(torch.autograd.Function):
    """gigl_gatv2_aggregate[_edge] / its backward over a CSR view (rowptr / rowend / col), before the bias; xe = the
    projected edge rows (lin_edge(edge_attr), `col` order) or None"""

    @staticmethod
    def forward(ctx, xl, xr, att, xe, eng, view, n_dev, heads, channels, slope):
        xl, xr, att = xl.contiguous(), xr.contiguous(), att.reshape(-1).contiguous()
        xe = xe.contiguous() if xe is not None else None
        out = eng.gatv2_aggregate(xl, xr, att, heads, channels, view, n_dev, None, negative_slope=slope, act=0,
                                  edge_rows=xe)
        ctx.save_for_backward(xl, xr, att, out, *([xe] if xe is not None else []))
        ctx.meta = (eng, view, n_dev, heads, channels, slope)
        return out

    @staticmethod
    def backward(ctx, dout):
        xl, xr, att, out = ctx.saved_tensors[:4]
        xe = ctx.saved_tensors[4] if len(ctx.saved_tensors) > 4 else None
        eng, view, n_dev, heads, channels, slope = ctx.meta
        dxl, dxr, datt, dxe = eng.gatv2_aggregate_backward(xl, xr, att, heads, channels, view, n_dev, out, dout,
                                                           negative_slope=slope, edge_rows=xe)
        return dxl, dxr, datt.view(1, heads, channels), dxe, None, None, None, None, None, None


class GATv2Conv(nn.Module):
    """parameter holder with PyG GATv2Conv's layout (lin_l / lin_r / att / bias; with edge_dim also lin_edge, no bias)"""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True,
                 negative_slope: float = 0.2, bias: bool = True, share_weights: bool = False,
                 edge_dim: Optional[int] = None, dropout: float = 0.0):
        super().__init__()
        if dropout:
            raise NotImplementedError("GATv2Conv attention dropout is not built")
        if not concat and heads > 1:
            raise NotImplementedError("GATv2Conv with averaged heads (concat=False, heads > 1) is not built")
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.negative_slope, self.share_weights, self.edge_dim = negative_slope, share_weights, edge_dim
        self.lin_l = nn.Linear(in_channels, heads * out_channels, bias=bias)
        self.lin_r = self.lin_l if share_weights else nn.Linear(in_channels, heads * out_channels, bias=bias)
        self.att = nn.Parameter(torch.empty(1, heads, out_channels))
        self.lin_edge = nn.Linear(edge_dim, heads * out_channels, bias=False) if edge_dim is not None else None
        self.bias = nn.Parameter(torch.zeros(heads * out_channels)) if bias else None
        nn.init.xavier_uniform_(self.lin_l.weight)
        nn.init.xavier_uniform_(self.lin_r.weight)
        nn.init.xavier_uniform_(self.att)
        if self.lin_edge is not None:
            nn.init.xavier_uniform_(self.lin_edge.weight)

    def forward(self, x: torch.Tensor, view, n_dev: torch.Tensor, eng: HipEngine,
                edge_attr: Optional[torch.Tensor] = None) -> torch.Tensor:
        xl = _linear(eng, x, self.lin_l.weight, self.lin_l.bias)
        xr = xl if self.share_weights else _linear(eng, x, self.lin_r.weight, self.lin_r.bias)
        xe = _linear(eng, edge_attr, self.lin_edge.weight, None) if self.lin_edge is not None else None
        out = _Gatv2AggFn.apply(xl, xr, self.att, xe, eng, view, n_dev, self.heads, self.out_channels, self.negative_slope)
        return out + self.bias if self.bias is not None else out


class GATv2(GraphSAGE):
    """conv_kwargs: heads, share_weights, negative_slope, bias, concat.  Head sizes the kernels are built for: channels a
    multiple of 4 with channels/4 a power of two <= 64, heads*channels <= 1024."""

    def __init__(self, in_dim: int, hid_dim: int, out_dim: int, num_layers: int = 2, edge_dim: Optional[int] = None,
                 **kwargs):
        ck = dict(kwargs.get("conv_kwargs") or {})
        for key in ("heads", "share_weights", "negative_slope", "bias", "concat"):
            if key in kwargs:
                ck[key] = kwargs.pop(key)
        kwargs.pop("conv_kwargs", None)
        heads = int(ck.get("heads", 1))
        super().__init__(in_dim, hid_dim, out_dim, num_layers=num_layers, **kwargs)
        last = hid_dim if (self.linear_layer or self.jk_layer is not None) else out_dim
        self.heads, self.edge_dim = heads, edge_dim
        self.conv_layers = nn.ModuleList([
            GATv2Conv(in_dim if i == 0 else hid_dim * heads, hid_dim if i < num_layers - 1 else last,
                      heads=heads if i < num_layers - 1 else 1, concat=bool(ck.get("concat", True)),
                      negative_slope=float(ck.get("negative_slope", 0.2)), bias=bool(ck.get("bias", True)),
                      share_weights=bool(ck.get("share_weights", False)), edge_dim=edge_dim)
            for i in range(num_layers)])
        if self.batchnorm:  # BatchNorm1d(hid_dim * num_heads), homogeneous.py:78-87
            self.batchnorm_layers = nn.ModuleList([nn.BatchNorm1d(hid_dim * heads)
                                                   for _ in range(len(self.batchnorm_layers))])

    def forward(self, batch, engine: Optional[HipEngine] = None) -> torch.Tensor:
        """GraphData -> every layer over the whole batch graph (autograd when grad mode is on): [n, out_dim]
        HipBatch  -> the trimmed schedule over the level-ordered union graph: [cap, out_dim], index with root_local"""
        from .models_attn import _CsrView
        from .nn import GraphData
        if isinstance(batch, GraphData):
            eng = engine or getattr(self, "engine", None)
            if eng is None:
                raise RuntimeError("GATv2.forward(GraphData) needs the HipEngine (model.engine = eng)")
            view = _CsrView(batch)
            ea = None
            if self.edge_dim is not None:
                if batch.edge_attr_csr is None:
                    raise ValueError(f"the model was built with edge_dim={self.edge_dim} but the batch has no edge features")
                ea = batch.edge_attr_csr
                if ea.shape[0] != view.col.numel():  # edgeless batch: col holds one padding entry
                    ea = torch.zeros((view.col.numel(), self.edge_dim), dtype=torch.float32, device=batch.x.device)
            h, xs = self._interact(batch.x, eng), []
            for l, conv in enumerate(self.conv_layers):
                h = self._post(conv(h, view, batch.n_dev, eng, ea), l, False)
                xs.append(h)
            if self.jk_layer is not None:
                h = self.jk_layer(xs)
            return self._head(h)
        with torch.no_grad():
            eng, u = batch.engine, batch.union
            L = self.num_layers
            assert u.hops == L, "one hop per layer"
            cap = int(u.nodes.numel())
            ea = None
            if self.edge_dim is not None:
                ea = batch.edge_attr if batch.edge_attr is not None else eng.union_edge_attr(u)
                assert ea.shape[1] == self.edge_dim
            batch = self._interacted(batch)
            if batch.x is None:
                h = eng.gather_rows(u.nodes, u.meta[0:1], cap)
            else:
                h = batch.x if batch.x_index is None else batch.x[batch.x_index.long()].contiguous()
            xs = []
            for l, conv in enumerate(self.conv_layers):
                # rows computed by layer l: the nodes that can still reach a root (a prefix of the level order); their
                # sources are the rows layer l-1 computed
                n_src = u.meta[GIGL_META_LEVEL0 + (L - l): GIGL_META_LEVEL0 + (L - l) + 1]
                n_dst = u.meta[GIGL_META_LEVEL0 + (L - 1 - l): GIGL_META_LEVEL0 + (L - l)]
                xl = eng.linear(h, conv.lin_l.weight.contiguous(), conv.lin_l.bias, n_src, cap, 0)
                xr = xl if conv.share_weights else eng.linear(h, conv.lin_r.weight.contiguous(), conv.lin_r.bias, n_dst,
                                                              cap, 0)
                xe = None
                if conv.lin_edge is not None:
                    xe = eng.linear(ea, conv.lin_edge.weight.contiguous(), None, dev_rows(ea), int(ea.shape[0]), 0)
                h = eng.gatv2_aggregate(xl, xr, conv.att.reshape(-1).contiguous(), conv.heads, conv.out_channels, u,
                                        n_dst, conv.bias, negative_slope=conv.negative_slope, act=0, edge_rows=xe)
                h = self._post(h, l, False).contiguous()
                xs.append(h)
            if self.jk_layer is not None:
                n_roots = int(u.meta[GIGL_META_LEVEL0].item())
                out = torch.zeros((cap, self.jk_layer.output_linear.out_features), dtype=torch.float32, device=h.device)
                out[:n_roots] = self.jk_layer([x[:n_roots] for x in xs])
                h = out
            return self._head(h)

    def make_plan(self, *args, **kwargs):
        raise NotImplementedError("the one-call plan computes GraphSAGE layers only; use forward(HipBatch)")


class DCNCross(nn.Module):
    """one cross layer of DCN-v2: out = x0 * (W x + b + diag_scale * x) + x, with W full (`_lin`) or low-rank
    (`_lin_v` after `_lin_u`, projection_dim wide) — the parameter names of the reference's layer
    (feature_interaction.py:7-101) so that its checkpoints load; the matrix products run on gigl_linear"""

    def __init__(self, in_dim: int, projection_dim: Optional[int] = None, diag_scale: float = 0.0, use_bias: bool = True):
        super().__init__()
        if diag_scale < 0.0:
            raise ValueError(f"diag_scale must not be negative (got {diag_scale})")
        self._in_dim, self._projection_dim, self._diag_scale, self._use_bias = in_dim, projection_dim, diag_scale, use_bias
        if projection_dim is None:
            self._lin = nn.Linear(in_dim, in_dim, bias=use_bias)
        else:
            self._lin_u = nn.Linear(in_dim, projection_dim, bias=use_bias)
            self._lin_v = nn.Linear(projection_dim, in_dim, bias=use_bias)

    def _product(self, eng: HipEngine, x: torch.Tensor) -> torch.Tensor:
        """W x + b through the full matrix or its two low-rank factors"""
        stages = [self._lin] if self._projection_dim is None else [self._lin_u, self._lin_v]
        for lin in stages:
            x = _linear(eng, x, lin.weight, lin.bias)
        return x

    def forward(self, x0: torch.Tensor, x: Optional[torch.Tensor] = None) -> torch.Tensor:
        from .models_hetero import _engine_for
        x = x0 if x is None else x
        if x0.shape[-1] != x.shape[-1]:
            raise ValueError(f"the base features ({x0.shape[-1]} wide) and the layer input ({x.shape[-1]} wide) must "
                             "have the same width")
        wx = self._product(_engine_for(self, x), x)
        if self._diag_scale:
            wx = wx + self._diag_scale * x
        return x0 * wx + x

    def reset_parameters(self):
        for m in self.children():
            m.reset_parameters()


class DCNv2(nn.Module):
    """num_layers stacked cross layers over the same x0 (feature_interaction.py:104-155)"""

    def __init__(self, in_dim: int, num_layers: int = 1, projection_dim: Optional[int] = None, diag_scale: float = 0.0,
                 use_bias: bool = True):
        super().__init__()
        self._in_dim, self._num_layers = in_dim, num_layers
        self._layers = nn.ModuleList([DCNCross(in_dim, projection_dim=projection_dim, diag_scale=diag_scale,
                                               use_bias=use_bias) for _ in range(num_layers)])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x0, xl = x, x
        for layer in self._layers:
            layer.engine = getattr(self, "engine", None)  # (the owning model's engine and stream, when it set one)
            xl = layer(x0, xl)
        return xl

    def reset_parameters(self):
        for layer in self._layers:
            layer.reset_parameters()
