"""fp32 CPU restatement of the reference's GNN forward for this path (TEST INFRASTRUCTURE ONLY).

The arithmetic lives in torch-geometric==2.5.3 (python/pyproject.toml:53), which is NOT under
/root/reference and not installable here, and no reference test pins conv outputs -> "parity
unpinned" (SURVEY.md §8(c)).  These functions restate PyG 2.5.3's documented formulas with plain
torch ops, in the reference's execution order: EVERY layer over the WHOLE batch union graph
(python/gigl/src/common/models/pyg/homogeneous.py:107-153).

  SAGEConv (aggr=mean, root_weight=True, bias=True):  out_i = W_l·mean_{j->i} x_j + b_l + W_r·x_i
  GCNConv  (add_self_loops, sym. norm):               out = D^-1/2 (A+I) D^-1/2 X W^T + b
  GATConv  (heads H, concat, negative_slope 0.2, add_self_loops): see gat_conv below
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def scatter_mean(src_rows: torch.Tensor, index: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros((n, src_rows.shape[1]), dtype=src_rows.dtype)
    out.index_add_(0, index, src_rows)
    cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src_rows.dtype)
    return out / cnt[:, None]


def sage_conv(x: torch.Tensor, edge_index: torch.Tensor, w_l: torch.Tensor, b_l: Optional[torch.Tensor],
              w_r: Optional[torch.Tensor]) -> torch.Tensor:
    src, dst = edge_index[0], edge_index[1]
    mean = scatter_mean(x[src], dst, x.shape[0])
    out = mean @ w_l.T
    if b_l is not None:
        out = out + b_l
    if w_r is not None:
        out = out + x @ w_r.T
    return out


def graphsage_forward(x: torch.Tensor, edge_index: torch.Tensor, state_dict, num_layers: int,
                      activation_after_last_conv: bool = False) -> torch.Tensor:
    """BasicHomogeneousGNN.forward with GraphSAGE convs, relu, no batchnorm/dropout (eval)"""
    h = x
    for i in range(num_layers):
        p = f"conv_layers.{i}."
        h = sage_conv(h, edge_index, state_dict[p + "lin_l.weight"], state_dict.get(p + "lin_l.bias"),
                      state_dict.get(p + "lin_r.weight"))
        if i < num_layers - 1 or activation_after_last_conv:
            h = torch.relu(h)
    return h


def gcn_conv(x: torch.Tensor, edge_index: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    n = x.shape[0]
    loops = torch.arange(n, dtype=edge_index.dtype)
    keep = edge_index[0] != edge_index[1]  # add_remaining_self_loops keeps one loop per node
    src = torch.cat([edge_index[0][keep], loops])
    dst = torch.cat([edge_index[1][keep], loops])
    deg = torch.zeros(n, dtype=x.dtype).index_add_(0, dst, torch.ones(dst.numel(), dtype=x.dtype))
    dinv = deg.pow(-0.5)
    dinv[torch.isinf(dinv)] = 0
    norm = dinv[src] * dinv[dst]
    xw = x @ w.T
    out = torch.zeros_like(xw).index_add_(0, dst, xw[src] * norm[:, None])
    return out + b if b is not None else out


def gat_conv(x: torch.Tensor, edge_index: torch.Tensor, w: torch.Tensor, att_src: torch.Tensor,
             att_dst: torch.Tensor, bias: Optional[torch.Tensor], heads: int, concat: bool = True,
             negative_slope: float = 0.2, edge_attr: Optional[torch.Tensor] = None,
             w_edge: Optional[torch.Tensor] = None, att_edge: Optional[torch.Tensor] = None,
             w_edge_msg: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PyG GATConv: h = xW viewed [N,H,C]; e_ij = leaky_relu(a_src·h_j + a_dst·h_i [+ a_edge·(W_e e_ij)]); softmax
    over the in-edges of i (self loops removed then added; with edge features the added loops carry the MEAN attribute
    of the node's remaining in-edges, add_self_loops(fill_value="mean")); out_i = sum_j alpha_ij h_j; concat or mean
    heads.  w_edge_msg (EdgeAttrGATConv.message, edge_attr_gat_conv.py:131-144): messages are h_j + W_msg e_ij."""
    n = x.shape[0]
    c = w.shape[0] // heads
    h = (x @ w.T).view(n, heads, c)
    loops = torch.arange(n, dtype=edge_index.dtype)
    keep = edge_index[0] != edge_index[1]
    src = torch.cat([edge_index[0][keep], loops])
    dst = torch.cat([edge_index[1][keep], loops])
    a_s = (h * att_src.view(1, heads, c)).sum(-1)
    a_d = (h * att_dst.view(1, heads, c)).sum(-1)
    logit = a_s[src] + a_d[dst]
    ea = None
    if edge_attr is not None:
        kept = edge_attr[keep]
        loop_attr = scatter_mean(kept, edge_index[1][keep], n)
        ea = torch.cat([kept, loop_attr])
        logit = logit + ((ea @ w_edge.T).view(-1, heads, c) * att_edge.view(1, heads, c)).sum(-1)
    e = torch.nn.functional.leaky_relu(logit, negative_slope)
    emax = torch.full((n, heads), float("-inf"), dtype=x.dtype).scatter_reduce(0, dst[:, None].expand(-1, heads), e,
                                                                              reduce="amax", include_self=True)
    ex = torch.exp(e - emax[dst])
    den = torch.zeros((n, heads), dtype=x.dtype).index_add_(0, dst, ex)
    alpha = ex / (den[dst] + 1e-16)
    msg = h[src]
    if ea is not None and w_edge_msg is not None:
        msg = msg + (ea @ w_edge_msg.T).view(-1, heads, c)
    out = torch.zeros((n, heads, c), dtype=x.dtype).index_add_(0, dst, msg * alpha[:, :, None])
    out = out.reshape(n, heads * c) if concat else out.mean(1)
    return out + bias if bias is not None else out


def retrieval_loss_rows(scores: torch.Tensor, query_ids: Sequence[int], candidate_ids: Sequence[int],
                        temperature: float = 0.07, remove_accidental_hits: bool = True) -> torch.Tensor:
    """Row-by-row restatement of RetrievalLoss (python/gigl/src/common/models/layers/loss.py:209-331) — pinned by
    the reference's known-answer tests (python/tests/unit/src/common/models/layers/loss_test.py:61-166, restated in
    tests/test_link_prediction.py).  Row i's target is column i.  Column j != i is masked out (logit = finfo.min)
    when it is another row of the same query (j < Q and query_ids[j] == query_ids[i]) or, with accidental-hit
    removal, when it holds the same candidate as the positive (candidate_ids[j] == candidate_ids[i]).
    Returns the SUM over rows of the softmax cross-entropy (CrossEntropyLoss(reduction="sum") against eye)."""
    q, c = scores.shape
    lo = torch.finfo(scores.dtype).min
    total = scores.new_zeros(())
    for i in range(q):
        row = scores[i] / temperature
        mask = torch.zeros(c, dtype=torch.bool)
        for j in range(c):
            if j == i:
                continue
            if j < q and query_ids[j] == query_ids[i]:
                mask[j] = True
            if remove_accidental_hits and candidate_ids[j] == candidate_ids[i]:
                mask[j] = True
        row = torch.where(mask, row + lo, row)
        total = total - torch.log_softmax(row, dim=0)[i]
    return total


def union_edge_index(rowptr, col) -> torch.Tensor:
    """CSR-by-destination (int arrays) -> PyG edge_index [2, E] (row 0 = src, row 1 = dst)"""
    rowptr = torch.as_tensor(rowptr, dtype=torch.int64)
    col = torch.as_tensor(col, dtype=torch.int64)
    deg = rowptr[1:] - rowptr[:-1]
    dst = torch.repeat_interleave(torch.arange(deg.numel(), dtype=torch.int64), deg)
    return torch.stack([col, dst])


# ---- heterogeneous convolutions (TEST INFRASTRUCTURE): edge-list restatements in plain torch ------------------------
def _segment_softmax(logits: torch.Tensor, index: torch.Tensor, n: int) -> torch.Tensor:
    """softmax of logits [E, H] within the groups given by index [E] (torch_geometric.utils.softmax: max-shifted,
    denominator + 1e-16)"""
    mx = torch.full((n, logits.shape[1]), float("-inf"), dtype=logits.dtype)
    mx = mx.scatter_reduce(0, index[:, None].expand_as(logits), logits, reduce="amax", include_self=True)
    ex = torch.exp(logits - mx[index])
    den = torch.zeros((n, logits.shape[1]), dtype=logits.dtype).index_add_(0, index, ex)
    return ex / (den[index] + 1e-16)


def hgt_conv(x_dict, edge_index_dict, p, heads: int):
    """HGTConv.forward (python/gigl/src/common/models/pyg/nn/conv/hgt_conv.py:161-244) on edge lists.
    p: dict with kqv[type] = (W [3F, in], b [3F]), out[type] = (W [F, F], b [F]), k_rel / v_rel [H*T, D, D] (type index
    h * T + t, :131-141), skip[type] scalar, p_rel[edge type] [H], edge_types (list: index t).  Every node type of
    x_dict comes back (types without in-edges aggregate zeros: the in-repo modification, :19-23)."""
    types = list(x_dict)
    F_ = p["out"][types[0]][0].shape[0]
    D = F_ // heads
    T = len(p["edge_types"])
    k, q, v = {}, {}, {}
    for t in types:
        w, b = p["kqv"][t]
        kqv = x_dict[t] @ w.T + b
        k[t], q[t], v[t] = (z.reshape(-1, heads, D) for z in torch.tensor_split(kqv, 3, dim=1))
    dst_off, c = {}, 0
    for t in types:
        dst_off[t] = c
        c += x_dict[t].shape[0]
    n_dst = c
    qq = torch.cat([q[t] for t in types])
    agg = torch.zeros((n_dst, heads, D), dtype=qq.dtype)
    logits, srcs_k, srcs_v, dsts = [], [], [], []
    for et, ei in edge_index_dict.items():
        ti = p["edge_types"].index(et)
        wk = torch.stack([p["k_rel"][h * T + ti] for h in range(heads)])  # [H, D, D]
        wv = torch.stack([p["v_rel"][h * T + ti] for h in range(heads)])
        kk = torch.einsum("nhd,hde->nhe", k[et[0]], wk)
        vv = torch.einsum("nhd,hde->nhe", v[et[0]], wv)
        d = ei[1] + dst_off[et[2]]
        a = (qq[d] * kk[ei[0]]).sum(-1) * p["p_rel"][et].reshape(1, heads) / (D ** 0.5)
        logits.append(a)
        srcs_v.append(vv[ei[0]])
        dsts.append(d)
    if logits:
        a = _segment_softmax(torch.cat(logits), torch.cat(dsts), n_dst)
        agg.index_add_(0, torch.cat(dsts), torch.cat(srcs_v) * a[:, :, None])
    out = {}
    for t in types:
        o = agg[dst_off[t]: dst_off[t] + x_dict[t].shape[0]].reshape(-1, F_)
        w, b = p["out"][t]
        o = torch.nn.functional.gelu(o) @ w.T + b
        if o.shape[1] == x_dict[t].shape[1]:
            al = torch.sigmoid(p["skip"][t])
            o = al * o + (1 - al) * x_dict[t]
        out[t] = o
    return out


def simplehgn_conv(edge_index, node_feat, edge_type, p, heads: int, out_dim: int, negative_slope: float = 0.2,
                   edge_feat=None):
    """SimpleHGNConv.forward (python/gigl/src/common/models/pyg/nn/conv/simplehgn_conv.py:113-180) on an edge list.
    Note the reference normalises alpha over the edges that share the SOURCE node (softmax(alpha, row), row =
    edge_index[0], :154-155) and sums the messages at the target.  p: W_nfeat [in, H*out], a_l / a_r [1, H, out],
    a_etype [1, H, Te], edge_type_emb [T, Te], W_etype (weight [T, Te, H*Te], bias [T, H*Te]), residual (W, b) | None,
    and with edge features W_efeat [Ein, H*Ein], a_efeat [1, H, Ein]."""
    n = node_feat.shape[0]
    emb = torch.nan_to_num(node_feat @ p["W_nfeat"], nan=0.0).reshape(n, heads, out_dim)
    row, col = edge_index[0], edge_index[1]
    te = p["edge_type_emb"].shape[1]
    w_et, b_et = p["W_etype"]
    et_vec = torch.stack([p["edge_type_emb"][t] @ w_et[t] + b_et[t] for t in range(w_et.shape[0])]).reshape(-1, heads, te)
    logit = (p["a_l"] * emb).sum(-1)[row] + (p["a_r"] * emb).sum(-1)[col] + (p["a_etype"] * et_vec).sum(-1)[edge_type]
    if edge_feat is not None:
        ein = edge_feat.shape[1]
        ee = torch.nan_to_num(edge_feat @ p["W_efeat"], nan=0.0).reshape(-1, heads, ein)
        logit = logit + (p["a_efeat"] * ee).sum(-1)
    alpha = _segment_softmax(torch.nn.functional.leaky_relu(logit, negative_slope), row, n)
    out = torch.zeros((n, heads, out_dim), dtype=emb.dtype).index_add_(0, col, emb[row] * alpha[:, :, None])
    out = out.reshape(n, heads * out_dim)
    if p.get("residual") is not None:
        w, b = p["residual"]
        out = out + node_feat @ w.T + b
    return out


def gin_conv(x: torch.Tensor, edge_index: torch.Tensor, w0: torch.Tensor, b0: torch.Tensor, w1: torch.Tensor,
             b1: torch.Tensor, eps: float = 0.0, act_first: bool = False, bn=None) -> torch.Tensor:
    """PyG 2.5.3 GINConv(nn=MLP([in, o, o], norm=batch_norm|None, plain_last)) as GIN.init_conv_layers builds it
    (python/gigl/src/common/models/pyg/homogeneous.py:205-249):  nn((1 + eps) x_i + sum_{j->i} x_j) with
    nn = lin0 -> [act] -> norm -> [act] -> lin1.   bn: (weight, bias, running_mean, running_var, eps) in eval mode."""
    src, dst = edge_index[0], edge_index[1]
    agg = torch.zeros_like(x)
    agg.index_add_(0, dst, x[src])
    h = (agg + (1.0 + eps) * x) @ w0.T + b0
    if act_first:
        h = torch.relu(h)
    if bn is not None:
        g, b, mu, var, e = bn
        h = (h - mu) / torch.sqrt(var + e) * g + b
    if not act_first:
        h = torch.relu(h)
    return h @ w1.T + b1


def transformer_conv(x: torch.Tensor, edge_index: torch.Tensor, p, heads: int, channels: int, concat: bool = True,
                     root_weight: bool = True, beta: bool = False, edge_attr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PyG 2.5.3 TransformerConv without edge features (Transformer.init_conv_layers, homogeneous.py:440-487):
    alpha_ij = softmax_j(<W_q x_i + b_q, W_k x_j + b_k> / sqrt(C)) over the in-edges of i (no self loops are added),
    out_i = sum_j alpha_ij (W_v x_j + b_v), heads concatenated or averaged, + lin_skip(x_i) (beta: gated).
    p: dict with lin_query/lin_key/lin_value/lin_skip .weight/.bias and lin_beta.weight"""
    n = x.shape[0]
    src, dst = edge_index[0], edge_index[1]
    q = (x @ p["lin_query.weight"].T + p["lin_query.bias"]).view(n, heads, channels)
    k = (x @ p["lin_key.weight"].T + p["lin_key.bias"]).view(n, heads, channels)
    v = (x @ p["lin_value.weight"].T + p["lin_value.bias"]).view(n, heads, channels)
    kj, vj = k[src], v[src]
    if edge_attr is not None:  # lin_edge(e) (no bias) joins the keys and the values of every edge
        e = (edge_attr @ p["lin_edge.weight"].T).view(-1, heads, channels)
        kj, vj = kj + e, vj + e
    logits = (q[dst] * kj).sum(-1) / (channels ** 0.5)
    alpha = _segment_softmax(logits, dst, n)
    out = torch.zeros((n, heads, channels), dtype=x.dtype)
    out.index_add_(0, dst, vj * alpha.unsqueeze(-1))
    out = out.reshape(n, heads * channels) if concat else out.mean(1)
    if root_weight:
        xr = x @ p["lin_skip.weight"].T
        if p.get("lin_skip.bias") is not None:
            xr = xr + p["lin_skip.bias"]
        if beta:
            b = torch.sigmoid(torch.cat([out, xr, out - xr], dim=-1) @ p["lin_beta.weight"].T)
            out = b * xr + (1 - b) * out
        else:
            out = out + xr
    return out


def gatv2_conv(x: torch.Tensor, edge_index: torch.Tensor, p, heads: int, channels: int, negative_slope: float = 0.2,
               share_weights: bool = False, edge_attr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PyG 2.5.3 GATv2Conv (GATv2.init_conv_layers, homogeneous.py:346-386): xl = lin_l(x), xr = lin_r(x) (= xl when
    share_weights); self loops removed then one added per node (with edge features: carrying the MEAN attribute of the
    node's in-edges, fill_value="mean"); z_ij = <att_h, leaky_relu(xl_j + xr_i [+ lin_edge(e_ij)])>; alpha = softmax over
    the in-edges of i; out_i = sum_j alpha_ij xl_j, heads concatenated, + bias.
    p: lin_l.weight/.bias, lin_r.weight/.bias, att [1, H, C], bias[, lin_edge.weight]"""
    n = x.shape[0]
    xl = (x @ p["lin_l.weight"].T + p["lin_l.bias"]).view(n, heads, channels)
    xr = xl if share_weights else (x @ p["lin_r.weight"].T + p["lin_r.bias"]).view(n, heads, channels)
    keep = edge_index[0] != edge_index[1]
    loops = torch.arange(n, dtype=edge_index.dtype)
    src = torch.cat([edge_index[0][keep], loops])
    dst = torch.cat([edge_index[1][keep], loops])
    s = xl[src] + xr[dst]
    if edge_attr is not None:
        ea = edge_attr[keep]
        mean = torch.zeros((n, ea.shape[1]), dtype=ea.dtype).index_add_(0, edge_index[1][keep], ea)
        cnt = torch.bincount(edge_index[1][keep], minlength=n).clamp(min=1).to(ea.dtype)
        ea = torch.cat([ea, mean / cnt[:, None]])
        s = s + (ea @ p["lin_edge.weight"].T).view(-1, heads, channels)
    s = torch.nn.functional.leaky_relu(s, negative_slope)
    alpha = _segment_softmax((s * p["att"].view(1, heads, channels)).sum(-1), dst, n)
    out = torch.zeros((n, heads, channels), dtype=x.dtype).index_add_(0, dst, xl[src] * alpha.unsqueeze(-1))
    return out.reshape(n, heads * channels) + p["bias"]


def gine_conv(x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor, w_e: torch.Tensor, b_e: torch.Tensor,
              w0: torch.Tensor, b0: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, eps: float = 0.0) -> torch.Tensor:
    """PyG 2.5.3 GINEConv(nn=MLP([in, o, o]), edge_dim) as GINE.init_conv_layers builds it (homogeneous.py:252-297):
    nn((1 + eps) x_i + sum_{j->i} relu(x_j + lin(e_ji))),  lin = Linear(edge_dim, in), nn = lin0 -> relu -> lin1"""
    src, dst = edge_index[0], edge_index[1]
    msg = torch.relu(x[src] + edge_attr @ w_e.T + b_e)
    agg = torch.zeros_like(x).index_add(0, dst, msg)
    h = torch.relu((agg + (1.0 + eps) * x) @ w0.T + b0)
    return h @ w1.T + b1
