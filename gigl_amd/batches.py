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


def build_batch_graph(samples: Sequence[Tuple[Sequence[wire.Node], Sequence[wire.Edge]]]):
    """-> (x [n,D] float32, edge_index [2,E] int64, global_to_local dict, node_order list of global ids)"""
    g2l: Dict[int, int] = {}
    feats: List[np.ndarray] = []
    es: List[int] = []
    ed: List[int] = []
    seen = set()
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
            key = (g2l[e.src_node_id], g2l[e.dst_node_id])
            if key in seen:
                continue
            seen.add(key)
            es.append(key[0])
            ed.append(key[1])
    n = len(feats)
    if n and feat_dim == 0:  # PygGraphBuilder: nodes without features get ones(1) (pyg_graph_builder.py:25-38)
        x = np.ones((n, 1), dtype=np.float32)
    else:
        x = np.stack(feats).astype(np.float32) if n else np.zeros((0, feat_dim or 0), np.float32)
    ei = np.array([es, ed], dtype=np.int64).reshape(2, -1)
    if ei.shape[1]:  # coalesce(): sort by (src, dst); duplicates were already dropped
        order = np.lexsort((ei[1], ei[0]))
        ei = ei[:, order]
    order_ids = [None] * n
    for g, l in g2l.items():
        order_ids[l] = g
    return x, ei, g2l, order_ids


@dataclass
class RootedNodeNeighborhoodBatch:
    graph: GraphData
    condensed_node_type_to_root_node_indices_map: Dict[int, torch.Tensor]
    root_nodes: List[Node]
    condensed_node_type_to_subgraph_id_to_global_node_id: Dict[int, Dict[int, int]]

    @staticmethod
    def collate_pyg_rooted_node_neighborhood_minibatch(samples: Sequence[wire.RootedNodeNeighborhood],
                                                       node_type: str = "node") -> "RootedNodeNeighborhoodBatch":
        x, ei, g2l, order = build_batch_graph([(s.neighborhood.nodes if s.neighborhood else [s.root_node],
                                                s.neighborhood.edges if s.neighborhood else []) for s in samples])
        roots = [Node(type=node_type, id=int(s.root_node.node_id)) for s in samples]
        idx = torch.tensor([g2l[r.id] for r in roots], dtype=torch.int64)
        return RootedNodeNeighborhoodBatch(
            graph=GraphData(x=torch.from_numpy(x), edge_index=torch.from_numpy(ei)),
            condensed_node_type_to_root_node_indices_map={0: idx}, root_nodes=roots,
            condensed_node_type_to_subgraph_id_to_global_node_id={0: {l: g for l, g in enumerate(order)}})

    @staticmethod
    def process_raw_pyg_samples_and_collate_fn(batch: Sequence[bytes], node_type: str = "node"):
        return RootedNodeNeighborhoodBatch.collate_pyg_rooted_node_neighborhood_minibatch(
            [wire.RootedNodeNeighborhood.FromString(b) for b in batch], node_type=node_type)


@dataclass
class SupervisedNodeClassificationBatch:
    graph: GraphData
    root_node_indices: torch.Tensor          # LongTensor[B]
    root_nodes: List[Node]
    root_node_labels: Optional[torch.Tensor]  # LongTensor[B]

    @staticmethod
    def collate_pyg_node_classification_minibatch(samples: Sequence[wire.SupervisedNodeClassificationSample],
                                                  node_type: str = "node") -> "SupervisedNodeClassificationBatch":
        x, ei, g2l, _ = build_batch_graph([(s.neighborhood.nodes, s.neighborhood.edges) for s in samples])
        roots = [Node(type=node_type, id=int(s.root_node.node_id)) for s in samples]
        labels = None
        if all(s.root_node_labels for s in samples) and samples:
            labels = torch.tensor([s.root_node_labels[0].label for s in samples], dtype=torch.int64)
        return SupervisedNodeClassificationBatch(
            graph=GraphData(x=torch.from_numpy(x), edge_index=torch.from_numpy(ei)),
            root_node_indices=torch.tensor([g2l[r.id] for r in roots], dtype=torch.int64), root_nodes=roots,
            root_node_labels=labels)

    @staticmethod
    def process_raw_pyg_samples_and_collate_fn(batch: Sequence[bytes], node_type: str = "node"):
        return SupervisedNodeClassificationBatch.collate_pyg_node_classification_minibatch(
            [wire.SupervisedNodeClassificationSample.FromString(b) for b in batch], node_type=node_type)


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
        x, ei, g2l, order = build_batch_graph([(s.neighborhood.nodes, s.neighborhood.edges) for s in samples])
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
            graph=GraphData(x=torch.from_numpy(x), edge_index=torch.from_numpy(ei)),
            root_node_indices=torch.tensor(roots, dtype=torch.int64),
            pos_supervision_edge_data={0: pos}, hard_neg_supervision_edge_data={0: neg},
            condensed_node_type_to_subgraph_id_to_global_node_id={0: {l: g for l, g in enumerate(order)}})

    @staticmethod
    def process_raw_pyg_samples_and_collate_fn(batch: Sequence[bytes]):
        return NodeAnchorBasedLinkPredictionBatch.collate_pyg_node_anchor_based_link_prediction_minibatch(
            [wire.NodeAnchorBasedLinkPredictionSample.FromString(b) for b in batch])


def iterate_tfrecord_batches(files: Sequence[str], batch_size: int, rank: int = 0, world_size: int = 1,
                             seed: int = 42, loop: bool = False):
    """TfRecordsIterableDataset (tf_records_iterable_dataset.py:49-82) + get_data_split_for_current_worker
    (data_loaders/utils.py:23-56): the file list is permuted once with RandomState(seed=42), files are strided
    across ranks, records are batched in order; loop=True cycles forever (LoopyIterableDataset :85-109)"""
    files = list(files)
    if files:
        files = list(np.random.RandomState(seed).permutation(np.array(files, dtype=object)))
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
