"""Batch dataclasses + collate for samples that arrive over the wire (TFRecords of protos).

Mirror of (paths relative to the reference root):
  RootedNodeNeighborhoodBatch        python/gigl/src/training/v1/lib/data_loaders/rooted_node_neighborhood_data_loader.py:35-45
    .collate_pyg_rooted_node_neighborhood_minibatch  :78-158,  .process_raw_pyg_samples_and_collate_fn :161-241
  SupervisedNodeClassificationBatch  .../supervised_node_classification_data_loader.py:32-41, collate :74-117,120-172
  GraphBuilder remap / dedup         python/gigl/src/common/graph_builder/abstract_graph_builder.py:16-24,49-150
  coalesce + to_homogeneous          .../data_loaders/utils.py:59-146
Semantics kept: global->local ids in first-seen order (nodes of a sample first, then its edges), an edge is
skipped if its (local src, local dst) is already present, edges are finally sorted by (src, dst), an empty
batch graph has edge_index of shape (2, 0).  Host side (numpy), like the reference's Python; the tensors are
then moved to the device where all message passing runs on the HIP kernels (gigl_amd.nn).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import wire
from .nn import GraphData


@dataclass(frozen=True)
class Node:
    """python/gigl/src/common/types/graph_data.py:39-51 (homogeneous: one node type)"""
    type: str
    id: int


def _opt_tensor(a):
    return None if a is None else torch.from_numpy(a)


def build_batch_graph(samples: Sequence[Tuple[Sequence[wire.Node], Sequence[wire.Edge]]]):
    """-> (x [n,D] float32, edge_index [2,E] int64, global_to_local dict, node_order list of global ids,
    edge_attr [E,De] float32 in edge_index order or None when the samples' edges carry no features)"""
    g2l: Dict[int, int] = {}
    feats: List[np.ndarray] = []
    es: List[int] = []
    ed: List[int] = []
    seen = set()
    efeats: List[np.ndarray] = []
    edge_dim: Optional[int] = None  # GraphBuilder.should_register_edge_features: fixed by the first edge added
    feat_dim: Optional[int] = None
    for nodes, edges in samples:
        for nd in nodes:
            if nd.node_id in g2l:
                if not np.allclose(feats[g2l[nd.node_id]], nd.feature_values):  # abstract_graph_builder.py:66-86
                    raise AssertionError(f"node {nd.node_id} re-added with different features")
                continue
            if feat_dim is None:
                feat_dim = int(np.asarray(nd.feature_values).size)
            g2l[nd.node_id] = len(feats)
            feats.append(np.asarray(nd.feature_values, dtype=np.float32))
        for e in edges:
            if e.src_node_id not in g2l or e.dst_node_id not in g2l:  # :26-30
                raise TypeError(f"Tried to fetch a node which we have no information on (edge "
                                f"{e.src_node_id}->{e.dst_node_id})")
            fv = np.asarray(e.feature_values, dtype=np.float32)
            if edge_dim is None:
                edge_dim = int(fv.size)
            if int(fv.size) != edge_dim:  # abstract_graph_builder.py:121-132
                raise TypeError(f"edge feature registration is inconsistent: edge {e.src_node_id}->{e.dst_node_id} "
                                "differs from the first edge")
            key = (g2l[e.src_node_id], g2l[e.dst_node_id])
            if key in seen:
                continue
            seen.add(key)
            es.append(key[0])
            ed.append(key[1])
            efeats.append(fv)
    n = len(feats)
    if n and feat_dim == 0:  # PygGraphBuilder: nodes without features get ones(1) (pyg_graph_builder.py:25-38)
        x = np.ones((n, 1), dtype=np.float32)
    else:
        x = np.stack(feats).astype(np.float32) if n else np.zeros((0, feat_dim or 0), np.float32)
    ei = np.array([es, ed], dtype=np.int64).reshape(2, -1)
    ea = np.stack(efeats).astype(np.float32) if edge_dim and efeats else None
    if ei.shape[1]:  # coalesce(): sort by (src, dst); duplicates were already dropped
        order = np.lexsort((ei[1], ei[0]))
        ei = ei[:, order]
        if ea is not None:
            ea = ea[order]
    order_ids = [None] * n
    for g, l in g2l.items():
        order_ids[l] = g
    return x, ei, g2l, order_ids, ea


def collate_serialized(batch: Sequence[bytes], kind: int, n_threads: int = 0):
    """native collate of serialized samples (libgigl_hip.so gigl_collate_records, host C++: parallel proto parsing,
    first-seen numbering, edge dedup, coalesce).  -> dict(x, edge_index, node_ids, root_local, labels, has_label,
    pos_off, pos_dst, neg_off, neg_dst) as numpy arrays.  Raises what the reference's Python loops raise:
    AssertionError (node re-added with different features), TypeError (edge endpoint unknown), KeyError (root or
    supervision target not in the batch graph), ValueError (malformed record)."""
    import ctypes as C
    import os

    from . import _lib
    lib = _lib.load()
    payloads = [bytes(b) for b in batch]
    lens = np.array([len(b) for b in payloads], dtype=np.int64)
    off = np.zeros(len(payloads), dtype=np.int64)
    if len(payloads) > 1:
        np.cumsum(lens[:-1], out=off[1:])
    blob = b"".join(payloads) or b"\0"
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = lib.gigl_collate_records(blob, C.c_void_p(off.ctypes.data), C.c_void_p(lens.ctypes.data), len(payloads), kind,
                                  n_threads or min(16, os.cpu_count() or 1), C.byref(h), err, 512)
    if rc != 0:
        msg = err.value.decode("utf-8", "replace")
        exc = (AssertionError if "re-added" in msg else TypeError if ("Tried to fetch" in msg or "edge feature" in msg)
               else KeyError if "not in the batch graph" in msg else ValueError)
        raise exc(msg or "gigl_collate_records failed")
    try:
        n, e, d, npos, nneg = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int64(), C.c_int64()
        lib.gigl_collated_info(h, C.byref(n), C.byref(e), C.byref(d), C.byref(npos), C.byref(nneg))
        b = len(payloads)
        out = dict(node_ids=np.empty(n.value, np.uint32), x=np.empty((n.value, d.value), np.float32),
                   edge_index=np.empty((2, e.value), np.int64), root_local=np.empty(b, np.int64),
                   labels=np.empty(b, np.int64), has_label=np.empty(b, np.uint8), pos_off=np.empty(b + 1, np.int64),
                   pos_dst=np.empty(npos.value, np.int64), neg_off=np.empty(b + 1, np.int64),
                   neg_dst=np.empty(nneg.value, np.int64))
        ptr = lambda a: C.c_void_p(a.ctypes.data)
        lib.gigl_collated_copy(h, ptr(out["node_ids"]), ptr(out["x"]), ptr(out["edge_index"]), ptr(out["root_local"]),
                               ptr(out["labels"]), ptr(out["has_label"]), ptr(out["pos_off"]), ptr(out["pos_dst"]),
                               ptr(out["neg_off"]), ptr(out["neg_dst"]))
        de = C.c_int32()
        lib.gigl_collated_edge_attr(h, C.byref(de), None)
        out["edge_attr"] = None
        if de.value:
            out["edge_attr"] = np.empty((e.value, de.value), np.float32)
            lib.gigl_collated_edge_attr(h, C.byref(de), ptr(out["edge_attr"]))
    finally:
        lib.gigl_collated_destroy(h)
    if n.value and d.value == 0:  # PygGraphBuilder: nodes without features get ones(1) (pyg_graph_builder.py:25-38)
        out["x"] = np.ones((n.value, 1), dtype=np.float32)
    return out


def collate_serialized_typed(batch: Sequence[bytes], kind: int, n_node_types: int,
                             edge_type_endpoints: Sequence[Tuple[int, int]], n_threads: int = 0):
    """native collate of serialized HETEROGENEOUS samples (gigl_collate_typed_records): nodes carry
    condensed_node_type, edges condensed_edge_type; `edge_type_endpoints[t]` = (condensed src node type, condensed dst
    node type) of condensed edge type t (GraphMetadataPbWrapper.condensed_edge_type_to_edge_type_map).
    -> dict(node_ids={nt: uint32[n]}, x={nt: float32[n, d]}, edge_index={et: int64[2, e]}, edge_attr={et: float32[e, de]
    or None}, root_type, root_local, labels, has_label, pos_off, pos_dst, pos_type, neg_off, neg_dst, neg_type).
    Per type the reference's semantics: first-seen numbering (one counter per node type), edges de-duplicated and
    coalesced per edge type; the same exceptions as collate_serialized."""
    import ctypes as C
    import os

    from . import _lib
    lib = _lib.load()
    payloads = [bytes(b) for b in batch]
    lens = np.array([len(b) for b in payloads], dtype=np.int64)
    off = np.zeros(len(payloads), dtype=np.int64)
    if len(payloads) > 1:
        np.cumsum(lens[:-1], out=off[1:])
    blob = b"".join(payloads) or b"\0"
    n_et = len(edge_type_endpoints)
    e_src = np.array([p[0] for p in edge_type_endpoints] or [0], dtype=np.int32)
    e_dst = np.array([p[1] for p in edge_type_endpoints] or [0], dtype=np.int32)
    ptr = lambda a: C.c_void_p(a.ctypes.data)
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = lib.gigl_collate_typed_records(blob, ptr(off), ptr(lens), len(payloads), kind, n_node_types, n_et, ptr(e_src),
                                        ptr(e_dst), n_threads or min(16, os.cpu_count() or 1), C.byref(h), err, 512)
    if rc != 0:
        msg = err.value.decode("utf-8", "replace")
        exc = (AssertionError if "re-added" in msg else TypeError if ("Tried to fetch" in msg or "edge feature" in msg)
               else KeyError if "not in the batch graph" in msg else ValueError)
        raise exc(msg or "gigl_collate_typed_records failed")
    try:
        nn, nd = np.zeros(n_node_types, np.int64), np.zeros(n_node_types, np.int32)
        en, ed = np.zeros(max(n_et, 1), np.int64), np.zeros(max(n_et, 1), np.int32)
        npos, nneg = C.c_int64(), C.c_int64()
        lib.gigl_collated_typed_info(h, ptr(nn), ptr(nd), ptr(en), ptr(ed), C.byref(npos), C.byref(nneg))
        b = len(payloads)
        out = dict(node_ids={}, x={}, edge_index={}, edge_attr={})
        for t in range(n_node_types):
            ids = np.empty(int(nn[t]), np.uint32)
            x = np.empty((int(nn[t]), int(nd[t])), np.float32)
            lib.gigl_collated_typed_nodes(h, t, ptr(ids), ptr(x))
            if nn[t] and nd[t] == 0:  # PygGraphBuilder: nodes without features get ones(1)
                x = np.ones((int(nn[t]), 1), dtype=np.float32)
            out["node_ids"][t], out["x"][t] = ids, x
        for t in range(n_et):
            ei = np.empty((2, int(en[t])), np.int64)
            ea = np.empty((int(en[t]), int(ed[t])), np.float32) if ed[t] else None
            lib.gigl_collated_typed_edges(h, t, ptr(ei), ptr(ea) if ea is not None else None)
            out["edge_index"][t], out["edge_attr"][t] = ei, ea
        out.update(root_type=np.empty(b, np.int32), root_local=np.empty(b, np.int64), labels=np.empty(b, np.int64),
                   has_label=np.empty(b, np.uint8), pos_off=np.empty(b + 1, np.int64),
                   pos_dst=np.empty(npos.value, np.int64), pos_type=np.empty(npos.value, np.int32),
                   neg_off=np.empty(b + 1, np.int64), neg_dst=np.empty(nneg.value, np.int64),
                   neg_type=np.empty(nneg.value, np.int32))
        lib.gigl_collated_typed_samples(h, ptr(out["root_type"]), ptr(out["root_local"]), ptr(out["labels"]),
                                        ptr(out["has_label"]), ptr(out["pos_off"]), ptr(out["pos_dst"]),
                                        ptr(out["pos_type"]), ptr(out["neg_off"]), ptr(out["neg_dst"]),
                                        ptr(out["neg_type"]))
    finally:
        lib.gigl_collated_typed_destroy(h)
    return out


@dataclass
class HeteroRootedNodeNeighborhoodBatch:
    """RootedNodeNeighborhoodBatch of a heterogeneous job (rooted_node_neighborhood_data_loader.py:78-158 with a
    HeteroData graph): the typed samples of a batch collated natively into one graph per node / edge type.
    `graph` is a gigl_amd.models_hetero.HeteroGraphData (x_dict / edge_index_dict keyed by the metadata's names)."""
    graph: object
    condensed_node_type_to_root_node_indices_map: Dict[int, torch.Tensor]
    root_nodes: List[Tuple[int, int]]  # (condensed node type, global id) per sample
    condensed_node_type_to_subgraph_id_to_global_node_id: Dict[int, Dict[int, int]]

    @staticmethod
    def process_raw_pyg_samples_and_collate_fn(batch: Sequence[bytes], condensed_node_type_to_name: Dict[int, str],
                                               condensed_edge_type_to_triple: Dict[int, Tuple[str, str, str]]):
        """condensed_edge_type_to_triple[c] = (src node type, relation, dst node type): GraphMetadataPbWrapper's
        condensed_edge_type_to_edge_type_map"""
        from ._lib import REC_ROOTED_NODE_NEIGHBORHOOD
        from .models_hetero import HeteroGraphData
        name_to_cnt = {v: k for k, v in condensed_node_type_to_name.items()}
        n_nt = max(condensed_node_type_to_name) + 1
        n_et = max(condensed_edge_type_to_triple) + 1 if condensed_edge_type_to_triple else 0
        ends = [(0, 0)] * n_et
        for c, (s_, _, d_) in condensed_edge_type_to_triple.items():
            ends[c] = (name_to_cnt[s_], name_to_cnt[d_])
        out = collate_serialized_typed(batch, REC_ROOTED_NODE_NEIGHBORHOOD, n_nt, ends)
        x_dict = {condensed_node_type_to_name[t]: torch.from_numpy(out["x"][t]) for t in condensed_node_type_to_name
                  if out["node_ids"][t].size}
        ei, ea = {}, {}
        for c, triple in condensed_edge_type_to_triple.items():
            ei[tuple(triple)] = torch.from_numpy(out["edge_index"][c])
            if out["edge_attr"][c] is not None:
                ea[tuple(triple)] = torch.from_numpy(out["edge_attr"][c])
        roots_by_type: Dict[int, List[int]] = {}
        for t, l in zip(out["root_type"].tolist(), out["root_local"].tolist()):
            roots_by_type.setdefault(t, []).append(l)
        return HeteroRootedNodeNeighborhoodBatch(
            graph=HeteroGraphData(x_dict, ei, ea),
            condensed_node_type_to_root_node_indices_map={t: torch.tensor(v, dtype=torch.int64)
                                                          for t, v in roots_by_type.items()},
            root_nodes=[(int(t), int(out["node_ids"][t][l])) for t, l in zip(out["root_type"], out["root_local"])],
            condensed_node_type_to_subgraph_id_to_global_node_id={
                t: {i: int(g) for i, g in enumerate(out["node_ids"][t].tolist())} for t in condensed_node_type_to_name})


@dataclass
class HeteroNodeAnchorBasedLinkPredictionBatch:
    """NodeAnchorBasedLinkPredictionBatch of a heterogeneous job (node_anchor_based_link_prediction_data_loader.py with a
    HeteroData graph): typed training samples collated natively into one graph per node / edge type; the supervision
    (positive / hard-negative) targets of every root as LOCAL ids of the supervision edge type's destination node type."""
    graph: object                                   # models_hetero.HeteroGraphData
    root_condensed_node_type: int
    root_node_indices: torch.Tensor                 # int64 [B]: local ids inside the root node type
    pos_targets: Dict[int, List[torch.Tensor]]      # condensed supervision edge type -> per root int64 local dst ids
    hard_neg_targets: Dict[int, List[torch.Tensor]]
    condensed_node_type_to_subgraph_id_to_global_node_id: Dict[int, np.ndarray]  # local id -> global id, per node type

    @staticmethod
    def process_raw_pyg_samples_and_collate_fn(batch: Sequence[bytes], condensed_node_type_to_name: Dict[int, str],
                                               condensed_edge_type_to_triple: Dict[int, Tuple[str, str, str]]):
        from ._lib import REC_NODE_ANCHOR_LINK_PRED
        from .models_hetero import HeteroGraphData
        name_to_cnt = {v: k for k, v in condensed_node_type_to_name.items()}
        n_nt = max(condensed_node_type_to_name) + 1
        n_et = max(condensed_edge_type_to_triple) + 1
        ends = [(0, 0)] * n_et
        for c, (s_, _, d_) in condensed_edge_type_to_triple.items():
            ends[c] = (name_to_cnt[s_], name_to_cnt[d_])
        out = collate_serialized_typed(batch, REC_NODE_ANCHOR_LINK_PRED, n_nt, ends)
        x_dict = {condensed_node_type_to_name[t]: torch.from_numpy(out["x"][t]) for t in condensed_node_type_to_name
                  if out["node_ids"][t].size}
        ei, ea = {}, {}
        for c, triple in condensed_edge_type_to_triple.items():
            ei[tuple(triple)] = torch.from_numpy(out["edge_index"][c])
            if out["edge_attr"][c] is not None:
                ea[tuple(triple)] = torch.from_numpy(out["edge_attr"][c])
        root_types = set(out["root_type"].tolist())
        if len(root_types) > 1:
            raise ValueError(f"training samples of one batch must share their root node type, found {sorted(root_types)}")
        b = len(batch)

        def per_root(off, dst, typ):
            res: Dict[int, List[torch.Tensor]] = {}
            for c in sorted(set(typ.tolist())):
                res[c] = [torch.from_numpy(dst[off[i]:off[i + 1]][typ[off[i]:off[i + 1]] == c].astype(np.int64))
                          for i in range(b)]
            return res
        return HeteroNodeAnchorBasedLinkPredictionBatch(
            graph=HeteroGraphData(x_dict, ei, ea),
            root_condensed_node_type=int(out["root_type"][0]) if b else 0,
            root_node_indices=torch.from_numpy(out["root_local"].astype(np.int64)),
            pos_targets=per_root(out["pos_off"], out["pos_dst"], out["pos_type"]),
            hard_neg_targets=per_root(out["neg_off"], out["neg_dst"], out["neg_type"]),
            condensed_node_type_to_subgraph_id_to_global_node_id={t: out["node_ids"][t].astype(np.int64)
                                                                  for t in condensed_node_type_to_name})


@dataclass
class RootedNodeNeighborhoodBatch:
    graph: GraphData
    condensed_node_type_to_root_node_indices_map: Dict[int, torch.Tensor]
    root_nodes: List[Node]
    condensed_node_type_to_subgraph_id_to_global_node_id: Dict[int, Dict[int, int]]

    @staticmethod
    def collate_pyg_rooted_node_neighborhood_minibatch(samples: Sequence[wire.RootedNodeNeighborhood],
                                                       node_type: str = "node") -> "RootedNodeNeighborhoodBatch":
        x, ei, g2l, order, ea = build_batch_graph([(s.neighborhood.nodes if s.neighborhood else [s.root_node],
                                                s.neighborhood.edges if s.neighborhood else []) for s in samples])
        roots = [Node(type=node_type, id=int(s.root_node.node_id)) for s in samples]
        idx = torch.tensor([g2l[r.id] for r in roots], dtype=torch.int64)
        return RootedNodeNeighborhoodBatch(
            graph=GraphData(x=torch.from_numpy(x), edge_index=torch.from_numpy(ei), edge_attr=_opt_tensor(ea)),
            condensed_node_type_to_root_node_indices_map={0: idx}, root_nodes=roots,
            condensed_node_type_to_subgraph_id_to_global_node_id={0: {l: g for l, g in enumerate(order)}})

    @staticmethod
    def process_raw_pyg_samples_and_collate_fn(batch: Sequence[bytes], node_type: str = "node"):
        """serialized RootedNodeNeighborhood records -> batch (native collate; same result as decoding with
        wire.RootedNodeNeighborhood.FromString and calling collate_pyg_rooted_node_neighborhood_minibatch)"""
        from ._lib import REC_ROOTED_NODE_NEIGHBORHOOD
        c = collate_serialized(batch, REC_ROOTED_NODE_NEIGHBORHOOD)
        order = c["node_ids"].tolist()
        idx = torch.from_numpy(c["root_local"])
        return RootedNodeNeighborhoodBatch(
            graph=GraphData(x=torch.from_numpy(c["x"]), edge_index=torch.from_numpy(c["edge_index"]),
                            edge_attr=_opt_tensor(c["edge_attr"])),
            condensed_node_type_to_root_node_indices_map={0: idx},
            root_nodes=[Node(type=node_type, id=int(order[l])) for l in c["root_local"].tolist()],
            condensed_node_type_to_subgraph_id_to_global_node_id={0: {l: g for l, g in enumerate(order)}})


@dataclass
class SupervisedNodeClassificationBatch:
    graph: GraphData
    root_node_indices: torch.Tensor          # LongTensor[B]
    root_nodes: List[Node]
    root_node_labels: Optional[torch.Tensor]  # LongTensor[B]

    @staticmethod
    def collate_pyg_node_classification_minibatch(samples: Sequence[wire.SupervisedNodeClassificationSample],
                                                  node_type: str = "node") -> "SupervisedNodeClassificationBatch":
        x, ei, g2l, _, ea = build_batch_graph([(s.neighborhood.nodes, s.neighborhood.edges) for s in samples])
        roots = [Node(type=node_type, id=int(s.root_node.node_id)) for s in samples]
        labels = None
        if all(s.root_node_labels for s in samples) and samples:
            labels = torch.tensor([s.root_node_labels[0].label for s in samples], dtype=torch.int64)
        return SupervisedNodeClassificationBatch(
            graph=GraphData(x=torch.from_numpy(x), edge_index=torch.from_numpy(ei), edge_attr=_opt_tensor(ea)),
            root_node_indices=torch.tensor([g2l[r.id] for r in roots], dtype=torch.int64), root_nodes=roots,
            root_node_labels=labels)

    @staticmethod
    def process_raw_pyg_samples_and_collate_fn(batch: Sequence[bytes], node_type: str = "node"):
        """serialized SupervisedNodeClassificationSample records -> batch (native collate)"""
        from ._lib import REC_ROOTED_NODE_NEIGHBORHOOD
        c = collate_serialized(batch, REC_ROOTED_NODE_NEIGHBORHOOD)
        order = c["node_ids"]
        labels = torch.from_numpy(c["labels"]) if len(batch) and c["has_label"].all() else None
        return SupervisedNodeClassificationBatch(
            graph=GraphData(x=torch.from_numpy(c["x"]), edge_index=torch.from_numpy(c["edge_index"]),
                            edge_attr=_opt_tensor(c["edge_attr"])),
            root_node_indices=torch.from_numpy(c["root_local"]),
            root_nodes=[Node(type=node_type, id=int(order[l])) for l in c["root_local"].tolist()],
            root_node_labels=labels)


@dataclass
class BatchSupervisionEdgeData:
    """node_anchor_based_link_prediction_data_loader.py:39-47 (nested class there)"""
    root_node_to_target_node_id: Dict[int, torch.Tensor]
    label_edge_features: Optional[Dict[int, torch.Tensor]] = None


@dataclass
class NodeAnchorBasedLinkPredictionBatch:
    """node_anchor_based_link_prediction_data_loader.py:38-58; collate :61-224.  Homogeneous: one condensed
    edge type (0) and one condensed node type (0); supervision edges carry no label features (the sampler
    writes none — NodeAnchorBasedLinkPredictionTask.scala:146-312)."""
    BatchSupervisionEdgeData = BatchSupervisionEdgeData
    graph: GraphData
    root_node_indices: torch.Tensor
    pos_supervision_edge_data: Dict[int, BatchSupervisionEdgeData]
    hard_neg_supervision_edge_data: Dict[int, BatchSupervisionEdgeData]
    condensed_node_type_to_subgraph_id_to_global_node_id: Dict[int, Dict[int, int]]

    @staticmethod
    def collate_pyg_node_anchor_based_link_prediction_minibatch(
            samples: Sequence[wire.NodeAnchorBasedLinkPredictionSample]) -> "NodeAnchorBasedLinkPredictionBatch":
        x, ei, g2l, order, ea = build_batch_graph([(s.neighborhood.nodes, s.neighborhood.edges) for s in samples])
        pos = BatchSupervisionEdgeData(root_node_to_target_node_id={})
        neg = BatchSupervisionEdgeData(root_node_to_target_node_id={})
        roots: List[int] = []
        for s in samples:
            r = g2l[s.root_node.node_id]
            roots.append(r)
            # a positive / hard negative that is not in the neighbourhood is a KeyError, like node_mapping[...] (:186,:199)
            pos.root_node_to_target_node_id[r] = torch.tensor([g2l[e.dst_node_id] for e in s.pos_edges],
                                                              dtype=torch.int64)
            neg.root_node_to_target_node_id[r] = torch.tensor([g2l[e.dst_node_id] for e in s.hard_neg_edges],
                                                              dtype=torch.int64)
        return NodeAnchorBasedLinkPredictionBatch(
            graph=GraphData(x=torch.from_numpy(x), edge_index=torch.from_numpy(ei), edge_attr=_opt_tensor(ea)),
            root_node_indices=torch.tensor(roots, dtype=torch.int64),
            pos_supervision_edge_data={0: pos}, hard_neg_supervision_edge_data={0: neg},
            condensed_node_type_to_subgraph_id_to_global_node_id={0: {l: g for l, g in enumerate(order)}})

    @staticmethod
    def process_raw_pyg_samples_and_collate_fn(batch: Sequence[bytes]):
        """serialized NodeAnchorBasedLinkPredictionSample records -> batch (native collate)"""
        from ._lib import REC_NODE_ANCHOR_LINK_PRED
        c = collate_serialized(batch, REC_NODE_ANCHOR_LINK_PRED)
        pos = BatchSupervisionEdgeData(root_node_to_target_node_id={})
        neg = BatchSupervisionEdgeData(root_node_to_target_node_id={})
        po, pd, no, nd = c["pos_off"], torch.from_numpy(c["pos_dst"]), c["neg_off"], torch.from_numpy(c["neg_dst"])
        for i, r in enumerate(c["root_local"].tolist()):
            pos.root_node_to_target_node_id[r] = pd[po[i]:po[i + 1]].clone()
            neg.root_node_to_target_node_id[r] = nd[no[i]:no[i + 1]].clone()
        return NodeAnchorBasedLinkPredictionBatch(
            graph=GraphData(x=torch.from_numpy(c["x"]), edge_index=torch.from_numpy(c["edge_index"]),
                            edge_attr=_opt_tensor(c["edge_attr"])),
            root_node_indices=torch.from_numpy(c["root_local"]),
            pos_supervision_edge_data={0: pos}, hard_neg_supervision_edge_data={0: neg},
            condensed_node_type_to_subgraph_id_to_global_node_id={0: {l: g for l, g in enumerate(c["node_ids"].tolist())}})


def permuted_files(files: Sequence[str], seed: int = 42) -> List[str]:
    """the file list permuted once with RandomState(seed) (tf_records_iterable_dataset.py:66-68)"""
    files = list(files)
    if not files:
        return files
    return list(np.random.RandomState(seed).permutation(np.array(files, dtype=object)))


def iterate_tfrecord_batches(files: Sequence[str], batch_size: int, rank: int = 0, world_size: int = 1,
                             seed: int = 42, loop: bool = False):
    """TfRecordsIterableDataset (tf_records_iterable_dataset.py:49-82) + get_data_split_for_current_worker
    (data_loaders/utils.py:23-56): the file list is permuted once with RandomState(seed=42), files are strided
    across ranks, records are batched in order; loop=True cycles forever (LoopyIterableDataset :85-109).  With
    more ranks than files the list is tiled first (utils.py:38-47: data is replicated rather than leaving a rank
    without batches — a rank that skips the loop would leave the others waiting in the DDP gradient all-reduce)."""
    files = permuted_files(files, seed)
    if files and world_size > len(files):
        files = files * (-(-world_size // len(files)))
    mine = files[rank::world_size]
    while True:
        buf: List[bytes] = []
        for f in mine:
            for rec in wire.read_tfrecords(f):
                buf.append(rec)
                if len(buf) == batch_size:
                    yield buf
                    buf = []
        if buf:
            yield buf
        if not loop or not mine:
            return
