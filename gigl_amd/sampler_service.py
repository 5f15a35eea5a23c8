"""HipKHopSamplerService — the in-process sampler operator, mirror of the reference trait

    KHopSamplerService  (scala_spark35/common/src/main/scala/graphdb/KHopSamplerService.scala:10-33)
        setup() / teardown()
        getKHopSubgraphForRootNode(rootNode, samplingOpDag)   -> RootedNodeNeighborhood
        getKHopSubgraphForRootNodes(rootNodes, samplingOpDag) -> Seq[RootedNodeNeighborhood]
        samplePositiveEdgeNeighborhoods(rootNode, edgeType, numPositives, dag) -> (Seq[Edge], Seq[Graph])

and of the per-root assembly in SGSPureSparkV1Task.createSubgraph
(scala/subgraph_sampler/src/main/scala/libs/task/pureSpark/SGSPureSparkV1Task.scala:671-820):
  edges = hop-1 edges ++ hop-2 edges;  nodes = array_distinct(hop-1 ++ hop-2 nodes ++ [root]);
  every node/edge carries its condensed type and (hydrated) feature values.
Sampling runs on the GPU (HipEngine.sample_khop); this module only reshapes the tree layout into the
wire messages of gigl_amd.wire.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import wire
from ._lib import GIGL_INVALID

INVALID = GIGL_INVALID


def tree_to_edge_lists(roots: np.ndarray, fanouts: Sequence[int], nbr: Sequence[np.ndarray]):
    """per-root (src, dst) arrays in the reference's order (hop-1 edges, then hop-2 edges, ...)"""
    roots = np.asarray(roots, dtype=np.uint32)
    b = roots.size
    per_root_src: List[List[np.ndarray]] = [[] for _ in range(b)]
    per_root_dst: List[List[np.ndarray]] = [[] for _ in range(b)]
    parent = roots
    per_root = 1
    for k, f in enumerate(fanouts):
        f = int(f)
        per_root *= f
        a = np.asarray(nbr[k], dtype=np.uint32).reshape(b, per_root)
        dst = np.repeat(parent.reshape(b, per_root // f), f, axis=1)
        valid = a != INVALID
        for i in range(b):
            m = valid[i]
            if m.any():
                per_root_src[i].append(a[i][m])
                per_root_dst[i].append(dst[i][m])
        parent = a.reshape(-1)
    out = []
    for i in range(b):
        if per_root_src[i]:
            out.append((np.concatenate(per_root_src[i]), np.concatenate(per_root_dst[i])))
        else:
            out.append((np.zeros(0, np.uint32), np.zeros(0, np.uint32)))
    return out


def _distinct_in_order(ids: np.ndarray) -> np.ndarray:
    _, first = np.unique(ids, return_index=True)
    return ids[np.sort(first)]


def build_rooted_node_neighborhood(root: int, src: np.ndarray, dst: np.ndarray, features: Optional[np.ndarray],
                                   condensed_node_type: Optional[int] = 0, condensed_edge_type: Optional[int] = 0,
                                   edge_features=None) -> wire.RootedNodeNeighborhood:
    """createSubgraph's per-root row as a RootedNodeNeighborhood message"""
    # node order: sources of hop-1 edges, then of hop-2 edges (their dsts are already listed), root last
    ids = _distinct_in_order(np.concatenate([src, np.array([root], dtype=np.uint32)]).astype(np.uint32))

    def node(v: int) -> wire.Node:
        fv = features[v] if features is not None else wire._EMPTY_F32
        return wire.Node(node_id=int(v), condensed_node_type=condensed_node_type,
                         feature_values=np.asarray(fv, dtype=np.float32))

    edges = [wire.Edge(src_node_id=int(s), dst_node_id=int(d), condensed_edge_type=condensed_edge_type,
                       feature_values=(edge_features(int(s), int(d)) if edge_features else wire._EMPTY_F32))
             for s, d in zip(src.tolist(), dst.tolist())]
    return wire.RootedNodeNeighborhood(root_node=node(root),
                                       neighborhood=wire.Graph(nodes=[node(int(v)) for v in ids], edges=edges))


def validate_rooted_node_neighborhood(rnn: wire.RootedNodeNeighborhood) -> None:
    """TaskOutputValidator.validationHelper (scala/subgraph_sampler/src/main/scala/libs/task/
    TaskOutputValidator.scala:84-107): every edge endpoint must be one of the neighbourhood nodes"""
    ids = {n.node_id for n in rnn.neighborhood.nodes}
    for e in rnn.neighborhood.edges:
        if e.src_node_id not in ids or e.dst_node_id not in ids:
            raise RuntimeError(f"edge {e.src_node_id}->{e.dst_node_id} of root {rnn.root_node.node_id} "
                               "references a node missing from the neighborhood")


class HipKHopSamplerService:
    """one instance per worker/partition, like the reference; owns a HipEngine between setup/teardown"""

    def __init__(self, n_nodes: int, src, dst, features: Optional[np.ndarray], is_graph_directed: bool,
                 device: int = 0, sampling_seed: int = 42, keep_multi_edges: bool = True):
        # keep_multi_edges: a directed graph keeps repeated (src, dst) records, as the reference's directed path does
        # (SGSPureSparkV1Task.scala:337,442); switched off when edge features are hydrated (one row per distinct edge)
        self.keep_multi_edges = keep_multi_edges
        self._args = (n_nodes, src, dst, is_graph_directed)
        self.features = None if features is None else np.ascontiguousarray(features, dtype=np.float32)
        self.device = device
        self.sampling_seed = sampling_seed
        self.engine = None

    def setup(self) -> None:
        from .engine import HipEngine  # raises without the HIP library / a GPU: no CPU fallback
        n, src, dst, directed = self._args
        self.engine = HipEngine(self.device)
        multi = bool(directed and self.keep_multi_edges)
        self.engine.build_from_coo(n, src, dst, is_directed=directed, keep_multi_edges=multi)
        # out-edge graph for positives: CSR by source == CSC of the reversed edges
        self.engine.build_from_coo(n, dst, src, is_directed=directed, out_graph=True, keep_multi_edges=multi)
        if self.features is not None:
            self.engine.load_features(self.features)

    def teardown(self) -> None:
        if self.engine is not None:
            self.engine.close()
            self.engine = None

    def __enter__(self):
        self.setup()
        return self

    def __exit__(self, *exc):
        self.teardown()

    # ---- KHopSamplerService API
    def sample_trees(self, root_ids: Sequence[int], fanouts: Sequence[int]):
        roots = np.asarray(root_ids, dtype=np.int64).astype(np.uint32)
        tree = self.engine.sample_khop(roots, fanouts, sampling_seed=self.sampling_seed)
        nbr = [t.cpu().numpy().view(np.uint32) for t in tree.nbr]
        return roots, nbr

    def getKHopSubgraphForRootNodes(self, root_ids: Sequence[int], fanouts: Sequence[int],
                                    validate: bool = True) -> List[wire.RootedNodeNeighborhood]:
        if self.engine is None:
            raise RuntimeError("setup() must be called before sampling")
        roots, nbr = self.sample_trees(root_ids, fanouts)
        out = []
        for r, (s, d) in zip(roots.tolist(), tree_to_edge_lists(roots, fanouts, nbr)):
            rnn = build_rooted_node_neighborhood(r, s, d, self.features)
            if validate:
                validate_rooted_node_neighborhood(rnn)
            out.append(rnn)
        return out

    def getKHopSubgraphForRootNode(self, root_id: int, fanouts: Sequence[int]) -> wire.RootedNodeNeighborhood:
        return self.getKHopSubgraphForRootNodes([root_id], fanouts)[0]

    def samplePositiveEdgeNeighborhoods(self, root_id: int, num_positives: int, fanouts: Sequence[int]
                                        ) -> Tuple[List[wire.Edge], List[wire.Graph]]:
        """positives = out-neighbours of the root (counter 3: NodeAnchorBasedLinkPredictionBaseTask.scala:19-104),
        each with its own k-hop neighbourhood (lookupDstNodeNeighborhood :106-198 re-uses the positive's
        own rooted sample, i.e. the sample with the positive as root)"""
        pos, cnt = self.engine.sample_positives(np.array([root_id], dtype=np.uint32), num_positives,
                                                sampling_seed=self.sampling_seed)
        ids = pos.cpu().numpy().view(np.uint32)[: int(cnt.cpu()[0])]
        edges = [wire.Edge(src_node_id=int(root_id), dst_node_id=int(p), condensed_edge_type=0) for p in ids]
        graphs = [r.neighborhood for r in self.getKHopSubgraphForRootNodes(ids.tolist(), fanouts)] if ids.size else []
        return edges, graphs
