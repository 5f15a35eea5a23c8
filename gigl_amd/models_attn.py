"""GCN / GAT encoders on the HIP aggregation kernels (inference over the level-ordered union graph).

Mirror of (paths relative to the reference root):
  TwoLayerGCN  python/gigl/src/common/models/pyg/homogeneous.py:488-546   conv1 -> relu -> dropout -> conv2
               [-> L2 normalise]; PyG GCNConv params `conv{1,2}.lin.weight`, `conv{1,2}.bias`.
               The reference applies F.dropout(training=self.is_training) with the CONSTRUCTOR flag (default True)
               even in infer_batch; parity is defined for is_training=False, which is
               what this inference path computes (no dropout).
  GAT          :300-343 + BasicHomogeneousGNN.forward :107-153: GATConv(in, hid, heads=H) ... last layer heads=1,
               relu between layers; PyG 2.5.3 GATConv params `lin.weight`, `att_src`, `att_dst`, `bias`.
Parameter names come from the un-vendored PyG 2.5.3 ("parity unpinned"); the arithmetic is
checked against oracle/gnn_ref.py (tests/test_gpu_attn.py).
Trimmed schedule as in models.GraphSAGE: layer l computes only the prefix of rows that can still reach a root.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ._lib import GIGL_META_LEVEL0
from .models import HipBatch
from .engine import dev_i32


class GCNConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.xavier_uniform_(self.lin.weight)  # PyG: glorot weight, zero bias


class _LinearFn(torch.autograd.Function):
    """y = x @ w^T on the exact-fp32 MFMA kernel (gigl_linear), with its two backward GEMMs"""

    @staticmethod
    def forward(ctx, x, w, eng, n_dev):
        x = x.contiguous()
        ctx.eng, ctx.n_dev = eng, n_dev
        ctx.save_for_backward(x, w)
        return eng.linear(x, w.contiguous(), None, n_dev, int(x.shape[0]), act=0)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        eng, dy = ctx.eng, dy.contiguous()
        n_out = dev_i32(dy.device, dy.shape[1])
        dw, _ = eng.linear_weight_grad(dy, x, ctx.n_dev)  # dW = dy^T x, the rows as the inner dimension
        dx = eng.linear(dy, w.t().contiguous(), None, ctx.n_dev, int(x.shape[0]), 0) if ctx.needs_input_grad[0] else None
        return dx, dw, None, None


class _GcnAggFn(torch.autograd.Function):
    """a = A_hat h over a coalesced batch graph (gigl_gcn_aggregate: self loops removed, one added, symmetric
    normalisation with deg = 1 + #non-self in-edges).  Backward = A_hat^T: scale the incoming gradient by dinv of
    the destination, scatter it along the edges (gigl_gather_reduce_backward, sum), swap a stored self-loop edge for
    the added one, scale by dinv of the source."""

    @staticmethod
    def forward(ctx, h, eng, g, view):
        n = int(h.shape[0])
        ctx.eng, ctx.g = eng, g
        return eng.gcn_aggregate(h.contiguous(), int(h.shape[1]), None, view, g.n_dev, None, 0)[:n]

    @staticmethod
    def backward(ctx, da):
        eng, g = ctx.eng, ctx.g
        n, d = int(da.shape[0]), int(da.shape[1])
        dev = da.device
        rowlen = (g.rowptr[1:] - g.rowptr[:-1]).to(torch.int64)
        dst_of = torch.repeat_interleave(torch.arange(n, device=dev), rowlen)
        e = int(dst_of.numel())
        self_edges = torch.bincount(dst_of[g.col[:e].to(torch.int64) == dst_of], minlength=n) if e else rowlen * 0
        dinv = (1.0 + (rowlen - self_edges).to(torch.float32)).rsqrt()
        gs = (da * dinv[:, None]).contiguous()
        acc = torch.zeros((n, d), dtype=torch.float32, device=dev)
        # the kernel is the backward of the [reduce | self] operand of the SAGE layer: the self half gets zeros here
        eng.gather_mean_backward(torch.cat([gs, torch.zeros_like(gs)], dim=1).contiguous(), d, g.rowptr, None, g.col,
                                 g.n_dev, n, acc, aggr="sum")
        acc += gs * (1.0 - self_edges.to(torch.float32))[:, None]
        return acc * dinv[:, None], None, None, None


class TwoLayerGCN(nn.Module):
    def __init__(self, in_dim: int, out_dim: int, hid_dim: int = 16, is_training: bool = True,
                 should_l2_normalize_output: bool = False, **kwargs):
        super().__init__()
        self.is_training = is_training
        self.should_normalize = should_l2_normalize_output
        self.conv1 = GCNConv(in_dim, hid_dim, bias=bool(kwargs.get("bias", True)))
        self.conv2 = GCNConv(hid_dim, out_dim, bias=bool(kwargs.get("bias", True)))

    def forward(self, batch, engine=None) -> torch.Tensor:
        """HipBatch  -> [cap, out_dim]; rows [0, n_level0) = distinct roots (index with batch.root_local)
        GraphData -> [n, out_dim] over the whole coalesced batch graph, autograd-capable (the reference's default
                     node-classification model trains through this): conv1 -> relu -> dropout(p=0.5,
                     training=is_training) -> conv2 (homogeneous.py:488-546)"""
        from .nn import GraphData
        if isinstance(batch, GraphData):
            return self._forward_graph(batch, engine or getattr(self, "engine", None))
        with torch.no_grad():
            return self._forward_union(batch)

    def _forward_graph(self, g, eng) -> torch.Tensor:
        if eng is None:
            raise RuntimeError("TwoLayerGCN.forward(GraphData) needs the HipEngine (model.engine = eng)")
        view = _CsrView(g)
        h = g.x.contiguous()
        for k, conv in enumerate((self.conv1, self.conv2)):
            h = _GcnAggFn.apply(_LinearFn.apply(h, conv.lin.weight, eng, g.n_dev), eng, g, view)
            if conv.bias is not None:
                h = h + conv.bias
            if k == 0:
                h = torch.nn.functional.dropout(torch.relu(h), p=0.5, training=self.is_training)
        if self.should_normalize:
            h = torch.nn.functional.normalize(h, p=2, dim=1)
        return h

    def _forward_union(self, batch: HipBatch) -> torch.Tensor:
        eng, u = batch.engine, batch.union
        assert u.hops == 2, "TwoLayerGCN needs 2-hop samples"
        cap = int(u.nodes.numel())
        n1 = u.meta[GIGL_META_LEVEL0 + 1: GIGL_META_LEVEL0 + 2]
        n0 = u.meta[GIGL_META_LEVEL0: GIGL_META_LEVEL0 + 1]
        # (A_hat X) W == A_hat (X W): aggregate first so that the projection only touches the needed rows
        a1 = eng.gcn_aggregate(None, self.conv1.in_channels, u.nodes, u, n1, None, 0)
        h1 = eng.linear(a1, self.conv1.lin.weight.contiguous(), self.conv1.bias, n1, cap, act=1)
        a2 = eng.gcn_aggregate(h1, self.conv2.in_channels, None, u, n0, None, 0)
        out = eng.linear(a2, self.conv2.lin.weight.contiguous(), self.conv2.bias, n0, cap, act=0)
        if self.should_normalize:
            out = torch.nn.functional.normalize(out, p=2, dim=1)
        return out


class _GatConvFn(torch.autograd.Function):
    """out = GATConv(x) over ALL rows of a coalesced batch graph, before the activation; heads concatenated (or one
    head).  Forward: gigl_linear + gigl_gat_aggregate[_edge]; backward: gigl_gat_aggregate_backward for the edge-wise
    part, dense algebra (attention vectors, projection) through gigl_linear / small torch reductions."""

    @staticmethod
    def forward(ctx, x, w, att_src, att_dst, bias, v_att, w_msg, eng, view, n_dev, heads, channels, slope, edge_attr):
        n = int(x.shape[0])
        x = x.contiguous()
        # (view.meta = the number of SOURCE rows, n_dev = the rows to compute: the same for a whole batch graph, fewer
        # destination rows when only the first levels of a level-ordered graph are needed)
        xw = eng.linear(x, w.contiguous(), None, view.meta, n, act=0)
        kw = {} if edge_attr is None else dict(edge_attr=edge_attr, att_edge_folded=v_att.contiguous(),
                                               w_edge_msg=w_msg.contiguous() if w_msg is not None else None)
        out = eng.gat_aggregate(xw, att_src.reshape(-1).contiguous(), att_dst.reshape(-1).contiguous(), heads, channels,
                                view, n_dev, bias, concat=True, negative_slope=slope, act=0, **kw)
        ctx.eng, ctx.view, ctx.n_dev, ctx.dims, ctx.slope, ctx.edge_attr = eng, view, n_dev, (heads, channels), slope, edge_attr
        ctx.save_for_backward(x, w, xw, att_src, att_dst, bias if bias is not None else x.new_zeros(0), out,
                              v_att if v_att is not None else x.new_zeros(0),
                              w_msg if w_msg is not None else x.new_zeros(0))
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w, xw, att_src, att_dst, bias, out, v_att, w_msg = ctx.saved_tensors
        eng, view, n_dev, (heads, ch), slope, edge_attr = ctx.eng, ctx.view, ctx.n_dev, ctx.dims, ctx.slope, ctx.edge_attr
        n, hc = int(x.shape[0]), heads * ch
        dy = dy.contiguous()
        out_pre = (out - bias) if bias.numel() else out
        u_msg = None
        if w_msg.numel():  # EdgeAttrGATConv: u_i = W_msg^T g_i per head
            u_msg = torch.einsum("nhc,hck->nhk", dy.view(n, heads, ch), w_msg.view(heads, ch, -1)).contiguous()
        dh, ds, dd, dae, z = eng.gat_aggregate_backward(
            xw, att_src.reshape(-1).contiguous(), att_dst.reshape(-1).contiguous(), heads, ch, view, n_dev,
            out_pre.contiguous(), dy, negative_slope=slope, edge_attr=edge_attr,
            att_edge_folded=v_att.contiguous() if v_att.numel() else None, u_msg=u_msg)
        dw_msg = None
        if z is not None:
            dw_msg = torch.einsum("nhc,nhk->hck", dy.view(n, heads, ch), z).reshape(hc, -1)
        if hc <= 1024:  # one pass over the projected rows (gigl_gat_backward_epilogue), dh updated in place
            dxw, g_s, g_d = eng.gat_backward_epilogue(dh, ds, dd, xw, att_src, att_dst, heads, ch, view.meta)
            d_att_src, d_att_dst = g_s.view_as(att_src), g_d.view_as(att_dst)
        else:
            xw3 = xw.view(n, heads, ch)
            d_att_src = (ds.unsqueeze(-1) * xw3).sum(0).view_as(att_src)
            d_att_dst = (dd.unsqueeze(-1) * xw3).sum(0).view_as(att_dst)
            dxw = (dh.view(n, heads, ch) + ds.unsqueeze(-1) * att_src.view(1, heads, ch)
                   + dd.unsqueeze(-1) * att_dst.view(1, heads, ch)).reshape(n, hc).contiguous()
        dev = dy.device
        n_out = dev_i32(dev, hc)
        dw, _ = eng.linear_weight_grad(dxw, x, view.meta)                                   # dW = dxw^T x
        dx = eng.linear(dxw, w.t().contiguous(), None, view.meta, n, 0) if ctx.needs_input_grad[0] else None
        db = dy.sum(0) if bias.numel() else None
        dv = dae.t().mm(edge_attr) if dae is not None else None
        return dx, dw, d_att_src, d_att_dst, db, dv, dw_msg, None, None, None, None, None, None, None


class _GatInputAggFn(torch.autograd.Function):
    """z [H, rows, d] = the attention-weighted sums of the STORED feature rows for the first `rows` nodes of a batch graph
    built in HBM, as a function of the folded attention vectors u [2H, d] (gigl_gat_input_aggregate + its backward); the
    rows are inputs: no gradient"""

    @staticmethod
    def forward(ctx, u, table, g, n_rows_dev, rows, heads, slope):
        u = u.contiguous()
        z = table.gat_input_aggregate(u, heads, g.rowptr, g.rowptr[1:], g.col, g.node_ids, n_rows_dev, rows, slope)
        ctx.table, ctx.g, ctx.n_rows_dev, ctx.dims = table, g, n_rows_dev, (rows, heads, slope)
        ctx.save_for_backward(u)
        return z

    @staticmethod
    def backward(ctx, dz):
        (u,) = ctx.saved_tensors
        rows, heads, slope = ctx.dims
        g = ctx.g
        du = ctx.table.gat_input_aggregate_backward(u, heads, g.rowptr, g.rowptr[1:], g.col, g.node_ids, ctx.n_rows_dev,
                                                    rows, dz.contiguous(), slope)
        return du, None, None, None, None, None, None


class GATConv(nn.Module):
    """parameter holder with PyG GATConv's layout (lin / att_src / att_dst / bias; with edge_dim also lin_edge and
    att_edge)"""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True,
                 negative_slope: float = 0.2, bias: bool = True, edge_dim: Optional[int] = None):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.edge_dim = concat, negative_slope, edge_dim
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels if concat else out_channels)) if bias else None
        nn.init.xavier_uniform_(self.lin.weight)
        nn.init.xavier_uniform_(self.att_src)
        nn.init.xavier_uniform_(self.att_dst)
        if edge_dim is not None:
            self.lin_edge = nn.Linear(edge_dim, heads * out_channels, bias=False)
            self.att_edge = nn.Parameter(torch.empty(1, heads, out_channels))
            nn.init.xavier_uniform_(self.lin_edge.weight)
            nn.init.xavier_uniform_(self.att_edge)
        else:
            self.lin_edge = self.att_edge = None

    def folded_att_edge(self) -> torch.Tensor:
        """[heads, edge_dim]: <lin_edge(e), att_edge> == <e, lin_edge.weight^T att_edge> per head"""
        w = self.lin_edge.weight.view(self.heads, self.out_channels, self.edge_dim)
        return (w * self.att_edge.view(self.heads, self.out_channels, 1)).sum(1).contiguous()

    def edge_message_weight(self) -> Optional[torch.Tensor]:
        return None  # plain GATConv: edge features only enter the attention logits


class EdgeAttrGATConv(GATConv):
    """GATConv whose messages are h_j + W e_ij (python/gigl/src/common/models/pyg/nn/conv/edge_attr_gat_conv.py:11-144):
    W = lin_edge's weight when share_edge_att_message_weight, a separate lin_edge_message otherwise"""

    def __init__(self, *args, share_edge_att_message_weight: bool = True, **kwargs):
        super().__init__(*args, **kwargs)
        if self.edge_dim is not None and not share_edge_att_message_weight:
            self.lin_edge_message = nn.Linear(self.edge_dim, self.heads * self.out_channels, bias=False)
            nn.init.xavier_uniform_(self.lin_edge_message.weight)
        else:
            self.lin_edge_message = None

    def edge_message_weight(self) -> Optional[torch.Tensor]:
        if self.edge_dim is None:
            return None
        lin = self.lin_edge_message if self.lin_edge_message is not None else self.lin_edge
        return lin.weight.contiguous()


class GAT(nn.Module):
    """GAT / EdgeAttrGAT of python/gigl/src/common/models/pyg/homogeneous.py:300-343, :391-440: `edge_dim` switches the
    edge-feature terms on, `conv="edge_attr_gat"` selects EdgeAttrGATConv"""

    def __init__(self, in_dim: int, hid_dim: int, out_dim: int, num_layers: int = 2, heads: int = 1,
                 activation_after_last_conv: bool = False, should_l2_normalize_embedding_layer_output: bool = False,
                 edge_dim: Optional[int] = None, conv: str = "gat", **conv_kwargs):
        super().__init__()
        self.num_layers = num_layers
        self.edge_dim = edge_dim
        self.activation_after_last_conv = activation_after_last_conv
        self.should_l2_normalize_embedding_layer_output = should_l2_normalize_embedding_layer_output
        self.input_side_first_layer = True  # (False: always project first — the comparison knob of the tests)
        assert conv in ("gat", "edge_attr_gat")
        extra = {}
        if conv == "edge_attr_gat":
            extra["share_edge_att_message_weight"] = bool(conv_kwargs.get("share_edge_att_message_weight", True))
        cls = EdgeAttrGATConv if conv == "edge_attr_gat" else GATConv
        self.conv_layers = nn.ModuleList([
            cls(in_dim if i == 0 else hid_dim * heads, hid_dim if i < num_layers - 1 else out_dim,
                heads=heads if i < num_layers - 1 else 1, concat=bool(conv_kwargs.get("concat", True)),
                negative_slope=float(conv_kwargs.get("negative_slope", 0.2)), bias=bool(conv_kwargs.get("bias", True)),
                edge_dim=edge_dim, **extra)
            for i in range(num_layers)])

    def forward(self, batch, engine=None) -> torch.Tensor:
        """HipBatch  -> trimmed schedule over the level-ordered union graph: [cap, out]; index with batch.root_local
        GraphData -> every layer over the whole coalesced batch graph (samples that arrived as TFRecords; the
                     reference's execution order): [n, out]; edge features from GraphData.edge_attr"""
        from .nn import GraphData
        if isinstance(batch, GraphData):  # autograd-capable (training through the plugins); under no_grad: inference
            return self._forward_graph(batch, engine or getattr(self, "engine", None))
        with torch.no_grad():
            return self._forward_union(batch)

    def _forward_union(self, batch: HipBatch) -> torch.Tensor:
        eng, u = batch.engine, batch.union
        L = self.num_layers
        assert u.hops == L, "one hop per layer"
        cap = int(u.nodes.numel())
        edge_attr = None
        if self.edge_dim is not None:
            edge_attr = batch.edge_attr if getattr(batch, "edge_attr", None) is not None else eng.union_edge_attr(u)
            assert edge_attr.shape[1] == self.edge_dim
        h = None
        for l, conv in enumerate(self.conv_layers):
            # sources of layer l are the rows computed by layer l-1 (all union nodes for the first layer)
            n_src = u.meta[GIGL_META_LEVEL0 + (L - l): GIGL_META_LEVEL0 + (L - l) + 1]
            n_dst = u.meta[GIGL_META_LEVEL0 + (L - 1 - l): GIGL_META_LEVEL0 + (L - l)]
            if l == 0 and edge_attr is None and conv.concat and self.input_side_first_layer \
                    and conv.in_channels > conv.heads * conv.out_channels:
                # wide input rows: logits from folded attention vectors, projection after the aggregation
                act = 1 if (L > 1 or self.activation_after_last_conv) else 0
                if self.input_side_first_layer == "fused":  # one pass, logits on the fly (what the one-call plan runs)
                    h = eng.gat_input_layer_fused(u.nodes, conv.lin.weight.contiguous(),
                                                  conv.att_src.reshape(-1).contiguous(),
                                                  conv.att_dst.reshape(-1).contiguous(), conv.heads, conv.out_channels, u,
                                                  n_dst, conv.bias, negative_slope=conv.negative_slope, act=act)
                else:
                    h = eng.gat_input_layer(u.nodes, conv.lin.weight.contiguous(), conv.att_src.reshape(-1).contiguous(),
                                            conv.att_dst.reshape(-1).contiguous(), conv.heads, conv.out_channels, u, n_src,
                                            n_dst, conv.bias, negative_slope=conv.negative_slope, act=act)
                if h is not None:
                    continue
            x = eng.gather_rows(u.nodes, n_src, cap) if l == 0 else h
            h = self._layer(eng, conv, l, x, u, n_src, n_dst, cap, edge_attr)
        if self.should_l2_normalize_embedding_layer_output:
            h = torch.nn.functional.normalize(h, p=2, dim=1)
        return h

    def _layer(self, eng, conv, l, x, u, n_src, n_dst, cap, edge_attr):
        hw = eng.linear(x, conv.lin.weight.contiguous(), None, n_src, cap, act=0)
        act = 1 if (l < self.num_layers - 1 or self.activation_after_last_conv) else 0
        kw = {}
        if edge_attr is not None:
            kw = dict(edge_attr=edge_attr, att_edge_folded=conv.folded_att_edge(), w_edge_msg=conv.edge_message_weight())
        return eng.gat_aggregate(hw, conv.att_src.reshape(-1).contiguous(), conv.att_dst.reshape(-1).contiguous(),
                                 conv.heads, conv.out_channels, u, n_dst, conv.bias, concat=conv.concat,
                                 negative_slope=conv.negative_slope, act=act, **kw)

    def make_plan(self, eng, b: int, fanouts, groups: int = 1):
        """one-call pipeline (sample -> union -> this model's forward -> one row per root) for batches of `b` roots on
        `eng` (engine.GatPlan); weights are snapshotted — plan.set_weights(*model.plan_params()) after updates.  The
        first layer runs from the input side (gigl_gat_input_layer_fused): plain GATConv layers, heads concatenated."""
        from .engine import GatPlan
        assert len(fanouts) == self.num_layers, "one hop per layer"
        if self.edge_dim is not None or any(not c.concat and c.heads > 1 for c in self.conv_layers) or \
                self.should_l2_normalize_embedding_layer_output:
            raise NotImplementedError("the one-call plan computes plain GATConv layers (no edge features, concatenated "
                                      "heads, no output normalisation); use forward(HipBatch)")
        w, a_s, a_d, bs = self.plan_params()
        return GatPlan(eng, w, a_s, a_d, bs, [c.heads for c in self.conv_layers],
                       [c.out_channels for c in self.conv_layers], b, fanouts,
                       negative_slope=self.conv_layers[0].negative_slope, act_last=self.activation_after_last_conv,
                       groups=groups)

    def make_dist_plan(self, comm, b: int, fanouts, group_roots=None, max_window_end: int = -1, **kw):
        """the sharded one-call plan of this model on a hash-partitioned graph (dist.DistGatPlan): same layer
        restrictions as make_plan"""
        from .dist import DistGatPlan
        assert len(fanouts) == self.num_layers, "one hop per layer"
        if self.edge_dim is not None or any(not c.concat and c.heads > 1 for c in self.conv_layers) or \
                self.should_l2_normalize_embedding_layer_output:
            raise NotImplementedError("the sharded plan computes plain GATConv layers (no edge features, concatenated "
                                      "heads, no output normalisation)")
        w, a_s, a_d, bs = self.plan_params()
        return DistGatPlan(comm, w, a_s, a_d, bs, [c.heads for c in self.conv_layers],
                           [c.out_channels for c in self.conv_layers], b, fanouts,
                           negative_slope=self.conv_layers[0].negative_slope, act_last=self.activation_after_last_conv,
                           group_roots=group_roots, max_window_end=max_window_end, **kw)

    def plan_params(self):
        cs = self.conv_layers
        return ([c.lin.weight.detach() for c in cs], [c.att_src.detach() for c in cs], [c.att_dst.detach() for c in cs],
                [None if c.bias is None else c.bias.detach() for c in cs])

    def _forward_graph(self, g, eng) -> torch.Tensor:
        if eng is None:
            raise RuntimeError("GAT.forward(GraphData) needs the HipEngine (model.engine = eng)")
        assert g.rowptr is not None, "move the GraphData to the device first (GraphData.to)"
        n = g.num_nodes
        view = _CsrView(g)
        edge_attr = None
        if self.edge_dim is not None:
            if g.edge_attr_csr is None:
                raise ValueError(f"the model was built with edge_dim={self.edge_dim} but the batch has no edge features")
            edge_attr = g.edge_attr_csr
            assert edge_attr.shape[1] == self.edge_dim
            if edge_attr.shape[0] != view.col.numel():  # edgeless batch: col holds one padding entry
                edge_attr = torch.zeros((view.col.numel(), self.edge_dim), dtype=torch.float32, device=g.rowptr.device)
        # autograd path: grad mode on, module in training mode (model.eval() or torch.no_grad() select inference)
        train = torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters())
        if train and edge_attr is None and self._input_side_training_applies(g):
            return self._forward_graph_input_side(g, eng)
        h = g.features().contiguous()
        for l, conv in enumerate(self.conv_layers):
            if not train:
                with torch.no_grad():
                    h = self._layer(eng, conv, l, h, view, g.n_dev, g.n_dev, n, edge_attr)
                continue
            if not conv.concat and conv.heads > 1:
                raise NotImplementedError("training with heads averaged (concat=False) is not built")
            v_att = conv.folded_att_edge() if edge_attr is not None else None
            w_msg = conv.edge_message_weight() if edge_attr is not None else None
            h = _GatConvFn.apply(h, conv.lin.weight, conv.att_src, conv.att_dst, conv.bias, v_att, w_msg, eng, view,
                                 g.n_dev, conv.heads, conv.out_channels, conv.negative_slope, edge_attr)
            if l < self.num_layers - 1 or self.activation_after_last_conv:
                h = torch.relu(h)
        if self.should_l2_normalize_embedding_layer_output:
            h = torch.nn.functional.normalize(h, p=2, dim=1)
        return h


def _gat_input_side_training_applies(self, g) -> bool:
    """a two-layer GAT over a batch built in HBM (level-ordered nodes, the feature table at hand) whose stored rows are
    wider than the first layer's output: the first layer can run from the input side for the rows the second reads"""
    if not self.input_side_first_layer or self.num_layers != 2 or g.table is None or g.levels is None or \
            g.node_ids is None or len(g.levels) != 3:
        return False
    c0 = self.conv_layers[0]
    d = int(getattr(g.table, "feat_dim", 0))
    return (c0.concat and c0.edge_dim is None and d == c0.in_channels and d > c0.heads * c0.out_channels and d % 4 == 0
            and d <= 1024 and c0.heads in (1, 2, 4) and self.conv_layers[1].edge_dim is None)


def _gat_forward_graph_input_side(self, g, eng) -> torch.Tensor:
    """training forward over a batch built in HBM, the work cut to what the roots' rows depend on:
    layer 0 for the nodes of level <= 1 only, from the INPUT side — attention-weighted sums of the stored rows
    (_GatInputAggFn: every edge's logit <x_j, W_h^T att_h> formed from the row as it is read), then one projection per head
    over those sums — instead of projecting every source row of the batch graph first; layer 1 for the roots only.
    The same function of the parameters as the whole-graph forward (rows of the roots; 1e-5), a fraction of its work:
    the batch graph of 1,024 anchors has ~330 k nodes, ~53 k of level <= 1."""
    n0, n1, n = (int(v) for v in g.levels)
    dev = g.rowptr.device
    c0, c1 = self.conv_layers
    H, C, d = c0.heads, c0.out_channels, c0.in_channels
    n1_dev = torch.tensor([n1], dtype=torch.int32, device=dev)
    n0_dev = torch.tensor([n0], dtype=torch.int32, device=dev)
    w3 = c0.lin.weight.view(H, C, d)
    u = torch.cat([torch.einsum("hcd,hc->hd", w3, c0.att_src.view(H, C)),
                   torch.einsum("hcd,hc->hd", w3, c0.att_dst.view(H, C))]).contiguous()      # [2H, d], differentiable
    z = _GatInputAggFn.apply(u, g.table, g, n1_dev, n1, H, c0.negative_slope)                  # [H, n1, d]
    h = torch.cat([_LinearFn.apply(z[k], w3[k], eng, n1_dev) for k in range(H)], dim=1)       # [n1, H*C]
    if c0.bias is not None:
        h = h + c0.bias
    h = torch.relu(h)
    view = _CsrView(g)
    view.meta, view.nodes = n1_dev, g.rowptr[1:n1 + 1]  # (sources: the n1 rows of h; only the length of `nodes` is read)
    h = _GatConvFn.apply(h, c1.lin.weight, c1.att_src, c1.att_dst, c1.bias, None, None, eng, view, n0_dev, c1.heads,
                         c1.out_channels, c1.negative_slope, None)[:n0]
    if self.activation_after_last_conv:
        h = torch.relu(h)
    if self.should_l2_normalize_embedding_layer_output:
        h = torch.nn.functional.normalize(h, p=2, dim=1)
    return h


class _CsrView:
    """the CSR of a coalesced GraphData in the shape gat_aggregate reads from a UnionGraph"""

    def __init__(self, g):
        self.rowptr, self.rowend, self.col = g.rowptr, g.rowptr[1:], g.col
        self.meta = g.n_dev           # meta[0] = number of nodes
        self.nodes = g.rowptr[1:]     # only its length (= node capacity) is read


GAT._input_side_training_applies = _gat_input_side_training_applies
GAT._forward_graph_input_side = _gat_forward_graph_input_side
