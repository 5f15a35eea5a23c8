"""SamplingOp-DAG sampler on the device — the heterogeneous `KHopSamplerService` of the reference's Spark-3.5 sampler.

Mirror of (paths relative to the reference root):
  SamplingOpDAG.from            scala_spark35/common/src/main/scala/types/SamplingOpDAG.scala:19-53
  GraphDBSampler                scala_spark35/subgraph_sampler/src/main/scala/libs/sampler/GraphDBSampler.scala:40-148
  LocalDbClient.executeQuery    scala_spark35/common/src/main/scala/graphdb/local/LocalDbClient.scala:156-237
  SamplingOp / SamplingDirection  proto/snapchat/research/gbml/subgraph_sampling_strategy.proto

Semantics kept: ops run breadth-first from the DAG's roots, an op runs once all its parents have; its input frontier for
a root node is the SET UNION of the node sets its parents returned for that root (the root itself for a root op); it
returns, per frontier node, up to `num_nodes_to_sample` neighbours along `edge_type` — INCOMING: sources of edges INTO
the frontier node, edge = (sampled -> frontier); OUTGOING: destinations, edge = (frontier -> sampled); the result is the
set union of every op's edges and nodes plus the root.  An op whose frontier is empty ends its path.

WHICH neighbours are taken is the one thing the reference leaves open (LocalDbClient keeps the first n of a Scala
HashSet, Nebula samples server-side; no seed reaches either) -> "parity unpinned" for the choice itself; here it is the
deterministic rule of the Spark sampler (SamplingStrategy.scala:16-82): permute the node's sorted neighbour list with
xxhash64 keys, K = root id + frontier node id, counter = 1 + the op's position in the DAG's op list.

Device work per op: one gigl_rows_dedup (frontier union, LDS hash per root) + one gigl_expand_frontier over B * width
slots on the op's edge-type graph; B roots advance together.  The typed RootedNodeNeighborhood /
NodeAnchorBasedLinkPredictionSample records (set union across ops, hydration with per-type node and edge features,
TFRecord framing) are written on the device too (gigl_typed_samples_encode: encode_records / encode_nablp_records); the
host assembly of the same messages (getKHopSubgraphForRootNodes / getNablpSamplesForRootNodes) is the surface the
reference's KHopSamplerService exposes and what the device bytes are checked against."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np
import torch

from . import wire

INCOMING, OUTGOING = "INCOMING", "OUTGOING"
INVALID = 0xFFFFFFFF


@dataclass(frozen=True)
class EdgeType:
    src_node_type: str
    relation: str
    dst_node_type: str


@dataclass
class SamplingOp:
    op_name: str
    edge_type: EdgeType
    num_nodes_to_sample: int
    input_op_names: Sequence[str] = ()
    sampling_direction: str = INCOMING  # proto default


@dataclass
class SamplingOpNode:
    sampling_op: SamplingOp
    parent_op_names: List[str] = field(default_factory=list)
    child_op_names: List[str] = field(default_factory=list)


class SamplingOpDAG:
    def __init__(self, nodes: Dict[str, SamplingOpNode], root_op_names: List[str], op_order: List[str]):
        self.nodes, self.root_op_names, self.op_order = nodes, root_op_names, op_order

    @classmethod
    def from_ops(cls, ops: Sequence[SamplingOp]) -> "SamplingOpDAG":
        nodes = {op.op_name: SamplingOpNode(op) for op in ops}
        if len(nodes) != len(ops):
            raise ValueError("sampling op names must be unique")
        roots = [op.op_name for op in ops if not op.input_op_names]
        for op in ops:
            for p in op.input_op_names:
                if p in nodes:  # unknown parents are dropped like filterKeys does (SamplingOpDAG.scala:44-45)
                    nodes[p].child_op_names.append(op.op_name)
                    nodes[op.op_name].parent_op_names.append(p)
        return cls(nodes, roots, [op.op_name for op in ops])

    def execution_order(self) -> List[str]:
        """the reference's queue discipline (GraphDBSampler.scala:54-127): dequeue, run when every parent has run
        (otherwise wait for the last parent to enqueue it again), enqueue the children"""
        done: List[str] = []
        queue = list(self.root_op_names)
        while queue:
            name = queue.pop(0)
            node = self.nodes[name]
            if name in done or any(p not in done for p in node.parent_op_names):
                continue
            done.append(name)
            queue.extend(node.child_op_names)
        return done


class SubgraphSamplingValidationError(ValueError):
    """a sampling-op DAG that cannot be traversed (python/gigl/src/common/types/exception.py
    SubgraphSamplingValidationErrorType: `error_type` carries the reference's category name)"""

    def __init__(self, error_type: str, message: str):
        super().__init__(f"{error_type}: {message}")
        self.error_type = error_type


def _frontier_type(op: SamplingOp) -> str:
    """node type an op starts from: INCOMING samples sources of edges INTO the frontier (frontier = dst type),
    OUTGOING destinations of edges out of it (frontier = src type)"""
    return op.edge_type.src_node_type if op.sampling_direction == OUTGOING else op.edge_type.dst_node_type


def _result_type(op: SamplingOp) -> str:
    return op.edge_type.dst_node_type if op.sampling_direction == OUTGOING else op.edge_type.src_node_type


def validate_sampling_op_dags(root_type_to_ops: Dict[str, Sequence[SamplingOp]], node_types: Sequence[str],
                              edge_types: Sequence[EdgeType], expected_root_node_types: Sequence[str] = ()) -> None:
    """the checks of SubgraphSamplingStrategyPbWrapper / SamplingOpPbWrapper (python/gigl/src/common/types/pb_wrappers/
    subgraph_sampling_strategy.py:26-260, sampling_op.py): unique op names, known input ops, no cycles, root node types
    known to the graph (and to the task, when given), a root op per non-empty DAG, edge types of the graph, and edge
    alignment — a root op starts from the DAG's root node type, a child op starts from the node type its parent returns.
    On typed graphs a misaligned edge would index one type's id space with another type's ids."""
    known_et = {(e.src_node_type, e.relation, e.dst_node_type) for e in edge_types}
    for root_type, ops in root_type_to_ops.items():
        if root_type not in node_types:
            raise SubgraphSamplingValidationError("ROOT_NODE_TYPE_NOT_IN_GRAPH_METADATA", f"root node type {root_type!r}")
        if expected_root_node_types and root_type not in expected_root_node_types:
            raise SubgraphSamplingValidationError("ROOT_NODE_TYPE_NOT_IN_TASK_METADATA", f"root node type {root_type!r}")
        by_name: Dict[str, SamplingOp] = {}
        for op in ops:
            if op.op_name in by_name:
                raise SubgraphSamplingValidationError("REPEATED_OP_NAME", f"op name {op.op_name!r} in the DAG of {root_type!r}")
            by_name[op.op_name] = op
        if ops and not any(not op.input_op_names for op in ops):
            raise SubgraphSamplingValidationError("MISSING_ROOT_SAMPLING_OP", f"the DAG of {root_type!r} has no op without inputs")
        for op in ops:
            et = op.edge_type
            if (et.src_node_type, et.relation, et.dst_node_type) not in known_et:
                raise SubgraphSamplingValidationError("SAMPLING_OP_EDGE_TYPE_NOT_IN_GRAPH_METADATA", f"{et} of op {op.op_name!r}")
            if not op.input_op_names and _frontier_type(op) != root_type:
                raise SubgraphSamplingValidationError(
                    "CONTAINS_INVALID_EDGE_IN_DAG", f"root op {op.op_name!r} ({op.sampling_direction}) starts from "
                    f"{_frontier_type(op)!r}, the DAG's root node type is {root_type!r}")
            for pname in op.input_op_names:
                if pname not in by_name:
                    raise SubgraphSamplingValidationError("BAD_INPUT_OP_NAME", f"op {op.op_name!r} names input {pname!r}")
                if _frontier_type(op) != _result_type(by_name[pname]):
                    raise SubgraphSamplingValidationError(
                        "CONTAINS_INVALID_EDGE_IN_DAG", f"op {op.op_name!r} starts from {_frontier_type(op)!r} but its "
                        f"input {pname!r} returns {_result_type(by_name[pname])!r}")
        state: Dict[str, int] = {}

        def visit(name: str) -> None:  # depth-first over the input edges: a node met again while open closes a cycle
            if state.get(name) == 1:
                raise SubgraphSamplingValidationError("DAG_CONTAINS_CYCLE", f"through op {name!r} in the DAG of {root_type!r}")
            if state.get(name) == 2:
                return
            state[name] = 1
            for pname in by_name[name].input_op_names:
                visit(pname)
            state[name] = 2
        for name in by_name:
            visit(name)
    for t in expected_root_node_types:
        if t not in root_type_to_ops:
            raise SubgraphSamplingValidationError("MISSING_EXPECTED_ROOT_NODE_TYPE", f"no DAG for node type {t!r}")


@dataclass
class OpResult:
    frontier: torch.Tensor  # [B, w] int32 (uint32 payload), INVALID = empty
    nbr: torch.Tensor       # [B, w, n]
    cnt: torch.Tensor       # [B, w]


class _DevMem:
    """library-owned device memory under the array interface torch.as_tensor understands (no copy)"""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _wrap(ptr, n: int, dtype, device) -> torch.Tensor:
    """a COPY (on the current stream) of n elements of library-owned device memory: the plan reuses its buffers"""
    n = int(n)
    if n <= 0 or not ptr:
        return torch.empty(0, dtype=dtype, device=device)
    typestr = {torch.int32: "<i4", torch.int64: "<i8"}[dtype]
    return torch.as_tensor(_DevMem(int(ptr), n, typestr), device=device).clone()


class HipGraphDBSampler:
    """typed graph resident in HBM (one CSR per edge type and direction) + the op-DAG executor"""

    def __init__(self, node_types: Dict[str, int], num_nodes: Dict[str, int],
                 edges: Dict[EdgeType, Tuple[np.ndarray, np.ndarray]], condensed_edge_types: Dict[EdgeType, int],
                 features: Optional[Dict[str, np.ndarray]] = None, device: int = 0, sampling_seed: int = 42,
                 edge_features: Optional[Dict[EdgeType, np.ndarray]] = None):
        from .engine import HipEngine
        self.node_types, self.num_nodes, self.condensed_edge_types = node_types, num_nodes, condensed_edge_types
        self.features = features or {}
        self.sampling_seed = sampling_seed
        self.engine = HipEngine(device)
        # Edge.feature_values of the typed messages (hydrated on the host with the message assembly): row i of
        # edge_features[et] belongs to edge i of edges[et]; several rows for one (src, dst): the first wins
        self._edge_rows: Dict[Tuple[int, int, int], np.ndarray] = {}
        self._has_edge_feats = {et: bool(np.asarray(rows).size) for et, rows in (edge_features or {}).items()}
        for et, rows in (edge_features or {}).items():
            c = condensed_edge_types[et]
            for s_, d_, row in zip(np.asarray(edges[et][0]).tolist(), np.asarray(edges[et][1]).tolist(),
                                   np.asarray(rows, dtype=np.float32)):
                self._edge_rows.setdefault((int(s_), int(d_), c), row)
        n_all = max(num_nodes.values())
        for et, (src, dst) in edges.items():
            src, dst = np.asarray(src), np.asarray(dst)
            # rows = the node a query starts from, columns = what it returns
            self.engine.load_label_edges(self._key(et, INCOMING), n_all, dst, src)
            # (the by-source list also carries the type's edge features for the device-side record encoder)
            self.engine.load_label_edges(self._key(et, OUTGOING), n_all, src, dst,
                                         feats=(edge_features or {}).get(et))

    @staticmethod
    def _key(et: EdgeType, direction: str) -> str:
        return f"{et.src_node_type}|{et.relation}|{et.dst_node_type}|{direction}"

    def close(self):
        for pl in getattr(self, "_plans", {}).values():  # one-call typed plans hold device buffers of their own
            try:
                self.engine._lib.gigl_typed_plan_destroy(pl["plan"])
            except Exception:  # noqa: BLE001 — closing must not raise
                pass
        self._plans = {}
        self.engine.close()

    def run_dag(self, roots: torch.Tensor, dag: SamplingOpDAG) -> Dict[str, OpResult]:
        eng = self.engine
        roots = roots.to(device=eng.device, dtype=torch.int32).contiguous()
        b = int(roots.numel())
        res: Dict[str, OpResult] = {}
        ran: Dict[str, torch.Tensor] = {}  # per root: did the op run (GraphDBSampler.scala:66-86)?
        for name in dag.execution_order():
            node = dag.nodes[name]
            op = node.sampling_op
            if not node.parent_op_names:
                front = roots.view(b, 1).clone()
                ran[name] = torch.ones(b, dtype=torch.bool, device=eng.device)
            else:
                front = torch.cat([res[p].nbr.view(b, -1) for p in node.parent_op_names], dim=1).contiguous()
                eng.rows_dedup(front)
                # an op runs for a root only when every parent ran for it and the united frontier is not empty;
                # a root for which it did not run contributes nothing and stops its children too
                ok = (front != -1).any(dim=1)
                for p in node.parent_op_names:
                    ok = ok & ran[p]
                ran[name] = ok
                front.masked_fill_(~ok.view(b, 1), -1)
            w, f = int(front.shape[1]), int(op.num_nodes_to_sample)
            ksum = (front + roots.view(b, 1)).contiguous()  # int32 add wraps like the sampler's K sums
            counter = 1 + dag.op_order.index(name)
            nbr, cnt = eng.expand_frontier(front.view(-1), ksum.view(-1), f, self.sampling_seed * counter, 1,
                                           label_edges=self._key(op.edge_type, op.sampling_direction))
            res[name] = OpResult(front, nbr.view(b, w, f), cnt.view(b, w))
        return res

    # ---- typed batch graph on the device (what the reference's collate + PygGraphBuilder produce from the typed
    #      RootedNodeNeighborhood samples of a batch: per node type the distinct nodes, per edge type the distinct
    #      edges, python/gigl/src/common/graph_builder/abstract_graph_builder.py:49-150) ---------------------------
    def batch_graph(self, root_ids: Sequence[int], root_node_type: str, dag: SamplingOpDAG):
        """-> (HeteroGraphData on the device, root_index int64 [B] into x_dict[root_node_type], {type: global ids}).
        The samples never leave HBM: the op results are merged there (ids of a type sorted ascending = its local
        numbering, edges distinct per edge type) and the rows of every type are gathered from its feature table."""
        from .models_hetero import HeteroGraphData
        eng = self.engine
        dev = eng.device
        roots = torch.tensor(np.asarray(root_ids, dtype=np.int64), device=dev)
        res = self.run_dag(roots.to(torch.int32), dag)
        graph, uniq = self._merge_dag_results([(dag, res)], {root_node_type: [roots]})
        with torch.cuda.stream(eng._stream):
            root_index = torch.searchsorted(uniq[root_node_type], roots)
        return graph, root_index, uniq

    def _merge_dag_results(self, parts, extra_ids: Dict[str, List[torch.Tensor]]):
        """the batch graph of several DAG runs (parts: [(dag, {op name: OpResult})]) plus `extra_ids` per node type (the
        roots): per type the distinct ids ascending = local numbering, per edge type the distinct edges, feature rows,
        edge attributes -> (HeteroGraphData, {type: global ids})"""
        from .models_hetero import HeteroGraphData
        eng = self.engine
        dev = eng.device
        ids: Dict[str, List[torch.Tensor]] = {t: list(v) for t, v in extra_ids.items()}
        pairs: Dict[EdgeType, List[torch.Tensor]] = {}
        with torch.cuda.stream(eng._stream):
            for dag, res in parts:
                for name, r in res.items():
                    op = dag.nodes[name].sampling_op
                    outgoing = op.sampling_direction == OUTGOING
                    et = op.edge_type
                    front_t = et.src_node_type if outgoing else et.dst_node_type
                    got_t = et.dst_node_type if outgoing else et.src_node_type
                    b, w, f = int(r.frontier.shape[0]), int(r.frontier.shape[1]), int(r.nbr.shape[2])
                    fr = (r.frontier.to(torch.int64) & 0xFFFFFFFF).view(b, w, 1).expand(b, w, f).reshape(-1)
                    nb = (r.nbr.to(torch.int64) & 0xFFFFFFFF).reshape(-1)
                    ok = (nb != INVALID) & (fr != INVALID)
                    fr, nb = fr[ok], nb[ok]
                    ids.setdefault(front_t, []).append(fr)
                    ids.setdefault(got_t, []).append(nb)
                    pairs.setdefault(et, []).append(torch.stack([fr, nb] if outgoing else [nb, fr]))
            uniq = {t: torch.unique(torch.cat(v)) for t, v in ids.items()}  # sorted: local id = rank
            x_dict = {}
            for t in sorted(uniq, key=lambda t_: self.node_types[t_]):
                u = uniq[t]
                tab = self._feature_table(t)
                x_dict[t] = tab.index_select(0, u) if tab is not None else torch.ones((u.numel(), 1), device=dev)
            ei = {}
            # (EVERY edge type of the graph, in condensed-type order, like the trainer-side collate's batches — a type
            # without edges in the batch keeps an empty list: an encoder that numbers the batch's edge types by their
            # order, SimpleHGN's to_homogeneous(), then sees the same numbering on both routes)
            for et, _c in sorted(self.condensed_edge_types.items(), key=lambda kv: kv[1]):
                ps = pairs.get(et)
                if ps is None or et.src_node_type not in uniq or et.dst_node_type not in uniq:
                    ei[(et.src_node_type, et.relation, et.dst_node_type)] = torch.zeros((2, 0), dtype=torch.int64, device=dev)
                    continue
                p2 = torch.cat(ps, dim=1)
                src = torch.searchsorted(uniq[et.src_node_type], p2[0])
                dst = torch.searchsorted(uniq[et.dst_node_type], p2[1])
                key = torch.unique(src * int(uniq[et.dst_node_type].numel()) + dst)
                nd = int(uniq[et.dst_node_type].numel())
                ei[(et.src_node_type, et.relation, et.dst_node_type)] = torch.stack([key // nd, key % nd])
            graph = HeteroGraphData(x_dict, ei)
            self._attach_edge_attr(graph, uniq)
        return graph, uniq

    def nablp_batch_graph(self, root_ids: Sequence[int], positive_edge_type: EdgeType, num_positives: int,
                          root_dag: SamplingOpDAG, positive_dag: SamplingOpDAG):
        """a typed link-prediction TRAINING batch in HBM: the batch graph the trainer-side collate builds from the typed
        NodeAnchorBasedLinkPredictionSamples of `root_ids` (each sample's neighbourhood = its root's merged with its
        positives': GraphDBNodeAnchorBasedLinkPredictionTask.scala:118-496, getNablpSamplesForRootNodes above) without the
        samples becoming records -> (graph, root_index int64 [b] into the source type, pos_local int64 [b, P] local ids in
        the destination type (-1: no such positive), {type: global ids})"""
        eng = self.engine
        roots, res, pos, pos_res = self._run_nablp(root_ids, positive_edge_type, num_positives, root_dag, positive_dag)
        b, P = len(root_ids), int(num_positives)
        with torch.cuda.stream(eng._stream):
            r64 = roots.to(torch.int64) & 0xFFFFFFFF
            p64 = (pos.nbr.to(torch.int64) & 0xFFFFFFFF).view(b, P)
            valid = p64 != INVALID
        extra = {positive_edge_type.src_node_type: [r64]}
        extra.setdefault(positive_edge_type.dst_node_type, []).append(p64[valid])
        graph, uniq = self._merge_dag_results([(root_dag, res), (positive_dag, pos_res)], extra)
        with torch.cuda.stream(eng._stream):
            root_index = torch.searchsorted(uniq[positive_edge_type.src_node_type], r64)
            u_dst = uniq[positive_edge_type.dst_node_type]
            loc = torch.searchsorted(u_dst, p64.clamp(max=int(u_dst[-1]) if u_dst.numel() else 0))
            pos_local = torch.where(valid, loc, torch.full_like(loc, -1))
        return graph, root_index, pos_local, uniq

    def _attach_edge_attr(self, graph, uniq: Dict[str, torch.Tensor]) -> None:
        """edge_attr_dict of a batch graph built in HBM: for every edge type that carries features, the feature row of
        each batch edge (Edge.feature_values of the typed samples, hydrateEdges of the sampler) — the edge's position
        in the type's resident CSR-by-source (binary search in the source's row: gigl_edge_ids) selects the row of the
        table load_label_edges laid out in that order.  What the trainer-side collate takes from the records' edge
        features (abstract_graph_builder.py:100-150: add_edge(feature_values))."""
        import ctypes as C
        from . import _lib
        eng = self.engine
        for et, has in self._has_edge_feats.items():
            key3 = (et.src_node_type, et.relation, et.dst_node_type)
            ei = graph.edge_index_dict.get(key3)
            if not has or ei is None or et.src_node_type not in uniq or et.dst_node_type not in uniq:
                continue
            entry = eng._label_edges[self._key(et, OUTGOING)]
            table = entry["table"]
            src_g = uniq[et.src_node_type][ei[0]].to(torch.int32).contiguous()
            dst_g = uniq[et.dst_node_type][ei[1]].to(torch.int32).contiguous()
            eid = torch.empty(int(src_g.numel()), dtype=torch.int64, device=eng.device)
            # (the by-source graph has rows = sources, columns = destinations: roles swapped, as load_label_edges builds it)
            _lib.check(eng._lib.gigl_edge_ids(eng._ctx, entry["graph"], C.c_void_p(dst_g.data_ptr()),
                                              C.c_void_p(src_g.data_ptr()), int(src_g.numel()), C.c_void_p(eid.data_ptr())),
                       eng._ctx)
            rows = table.index_select(0, eid.clamp(min=0))
            rows.masked_fill_((eid < 0).unsqueeze(1), 0.0)
            graph.edge_attr_dict[key3] = rows

    # ---- the same batch graph from the library's one-call plan (gigl_typed_plan_*, csrc/typed_plan.hip): the ops, the
    #      per-type distinct ids and the per-edge-type distinct edges are one stream of device work; the host reads
    #      the counts ONCE per batch (batch_graph above synchronises at every torch.unique / boolean mask)
    def typed_plan(self, root_node_type: str, dag: SamplingOpDAG, b_max: int):
        """-> the (cached) one-call plan for batches of up to b_max roots of `root_node_type` through `dag`"""
        import ctypes as C
        from . import _lib
        if not hasattr(self, "_plans"):
            self._plans: Dict[tuple, dict] = {}
        key = (root_node_type, id(dag), int(b_max))
        if key in self._plans:
            return self._plans[key]
        eng = self.engine
        order = dag.execution_order()
        index = {name: i for i, name in enumerate(order)}
        types = sorted(self.node_types, key=lambda t: self.node_types[t])
        tix = {t: i for i, t in enumerate(types)}
        slots: Dict[EdgeType, int] = {}
        ops = (_lib.GiglDagOp * len(order))()
        for i, name in enumerate(order):
            node = dag.nodes[name]
            op = node.sampling_op
            outgoing = op.sampling_direction == OUTGOING
            et = op.edge_type
            o = ops[i]
            o.graph = eng._label_edges[self._key(et, op.sampling_direction)]["graph"]
            o.fanout = int(op.num_nodes_to_sample)
            o.n_parents = len(node.parent_op_names)
            for k, pn in enumerate(node.parent_op_names):
                o.parents[k] = index[pn]
            o.hash_add = ((self.sampling_seed * (1 + dag.op_order.index(name)) + 2**31) % 2**32) - 2**31
            o.frontier_node_type = tix[et.src_node_type if outgoing else et.dst_node_type]
            o.result_node_type = tix[et.dst_node_type if outgoing else et.src_node_type]
            o.edge_slot = slots.setdefault(et, len(slots))
            o.outgoing = 1 if outgoing else 0
        plan = C.c_void_p()
        _lib.check(eng._lib.gigl_typed_plan_create(eng._ctx, ops, len(order), len(types), tix[root_node_type], len(slots),
                                                   int(b_max), C.byref(plan)), eng._ctx)
        out = _lib.GiglTypedPlanOut()
        _lib.check(eng._lib.gigl_typed_plan_buffers(plan, C.byref(out)), eng._ctx)
        entry = {"plan": plan, "out": out, "types": types, "slots": slots, "order": order, "b_max": int(b_max),
                 "ops": ops}
        self._plans[key] = entry
        return entry

    def batch_graph_plan(self, root_ids: Sequence[int], root_node_type: str, dag: SamplingOpDAG, b_max: int = 0,
                         edge_type_ids: Optional[Dict[tuple, int]] = None):
        """batch_graph through the one-call plan: identical results (same numbering: a type's distinct ids ascending,
        an edge type's distinct edges ascending by (src, dst)).
        edge_type_ids {(src type, relation, dst type): id} (a typed attention model's edge-type numbering, e.g.
        HGTConv.edge_types_map): the plan also merges the batch's edges into ONE CSR by destination
        (gigl_typed_plan_merged_csr) — graph.merged_csr, which HGT.forward takes instead of building it with torch ops"""
        return self.batch_graph_plan_finish(self.batch_graph_plan_issue(root_ids, root_node_type, dag, b_max, edge_type_ids))

    def batch_graph_plan_issue(self, root_ids: Sequence[int], root_node_type: str, dag: SamplingOpDAG, b_max: int = 0,
                               edge_type_ids: Optional[Dict[tuple, int]] = None) -> dict:
        """first half of batch_graph_plan: the batch's device work is enqueued (plan, merged CSR, the counts on their way
        to pinned host memory) and nothing is waited for.  A loop that issues batch i+1 BEFORE it launches the model over
        batch i never idles on the counts: `finish(i) -> issue(i+1) -> model(i)` (Inferencer's typed in-HBM route).
        One batch per plan may be in flight: finish a ticket before the next issue on the same (type, dag, b_max)."""
        import ctypes as C
        from . import _lib
        eng = self.engine
        dev = eng.device
        b = len(root_ids)
        pl = self.typed_plan(root_node_type, dag, max(int(b_max), b))
        out = pl["out"]
        roots64 = torch.from_numpy(np.asarray(root_ids, dtype=np.int64))
        roots = roots64.to(torch.int32).pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else roots64.to(torch.int32)
        nt, ns = len(pl["types"]), len(pl["slots"])
        ticket = {"pl": pl, "b": b, "root_type": root_node_type, "roots": roots, "edge_type_ids": edge_type_ids}
        with torch.cuda.stream(eng._stream):
            _lib.check(eng._lib.gigl_typed_plan_run(pl["plan"], C.c_void_p(roots.data_ptr()), b), eng._ctx)
            counts = torch.empty(nt + ns, dtype=torch.int32, device=dev)
            counts[:nt] = _wrap(out.n_nodes, nt, torch.int32, dev)
            counts[nt:] = _wrap(out.n_edges, ns, torch.int32, dev)
            ring = pl.setdefault("host_counts", [torch.empty(nt + ns, dtype=torch.int32).pin_memory() for _ in range(2)])
            pl["host_turn"] = (pl.get("host_turn", 0) + 1) % len(ring)
            host = ring[pl["host_turn"]]
            host.copy_(counts, non_blocking=True)  # the one host read of the batch
            ev = torch.cuda.Event()
            ev.record(eng._stream)
            ticket["counts"], ticket["event"] = host, ev
            if edge_type_ids is not None:
                used = [i for i, t in enumerate(pl["types"]) if int(out.nodes_cap[i]) > 0]
                by_type = sorted(pl["slots"], key=lambda et: self.condensed_edge_types[et])
                slot_ets = [(et.src_node_type, et.relation, et.dst_node_type) for et in by_type]
                t_arr = (C.c_int32 * len(used))(*used)
                s_arr = (C.c_int32 * len(slot_ets))(*[pl["slots"][et] for et in by_type])
                e_arr = (C.c_int32 * len(slot_ets))(*[int(edge_type_ids[k]) for k in slot_ets])
                csr = _lib.GiglTypedCsrOut()
                _lib.check(eng._lib.gigl_typed_plan_merged_csr(pl["plan"], b, t_arr, len(used), s_arr, e_arr, len(slot_ets),
                                                               C.byref(csr)), eng._ctx)
                ticket["csr"], ticket["used"], ticket["slot_ets"] = csr, used, slot_ets
        return ticket

    def batch_graph_plan_finish(self, ticket: dict):
        """second half: waits for the batch's counts and hands out its tensors (copies: the plan's buffers are
        overwritten by its next batch) -> (graph, root_index, distinct ids per type)"""
        from .models_hetero import HeteroGraphData
        eng = self.engine
        dev = eng.device
        pl, b, root_node_type = ticket["pl"], ticket["b"], ticket["root_type"]
        out = pl["out"]
        nt = len(pl["types"])
        ticket["event"].synchronize()
        h = ticket["counts"].tolist()
        with torch.cuda.stream(eng._stream):
            uniq, x_dict, ei = {}, {}, {}
            for i, t in enumerate(pl["types"]):
                n_t = int(h[i])
                if int(out.nodes_cap[i]) == 0:  # no op touches the type
                    continue
                u = _wrap(out.nodes[i], n_t, torch.int32, dev).to(torch.int64) & 0xFFFFFFFF
                uniq[t] = u
                tab = self._feature_table(t)
                x_dict[t] = tab.index_select(0, u) if tab is not None else torch.ones((n_t, 1), device=dev)
            for et, _c in sorted(self.condensed_edge_types.items(), key=lambda kv: kv[1]):
                sl = pl["slots"].get(et)
                if sl is None:  # (no op samples this edge type: an empty list, like the collate's batches)
                    ei[(et.src_node_type, et.relation, et.dst_node_type)] = torch.zeros((2, 0), dtype=torch.int64, device=dev)
                    continue
                n_e = int(h[nt + sl])
                keys = _wrap(out.edges[sl], n_e, torch.int64, dev)
                ei[(et.src_node_type, et.relation, et.dst_node_type)] = torch.stack([keys >> 32, keys & 0xFFFFFFFF])
            root_index = _wrap(out.root_index, b, torch.int32, dev).to(torch.int64)
            graph = HeteroGraphData(x_dict, ei)
            self._attach_edge_attr(graph, uniq)
            if ticket.get("csr") is not None:
                csr, used, slot_ets, edge_type_ids = ticket["csr"], ticket["used"], ticket["slot_ets"], ticket["edge_type_ids"]
                n_dst = sum(int(h[i]) for i in used)
                n_e = sum(int(h[nt + pl["slots"][et]]) for et in pl["slots"])
                take = lambda ptr, n: _wrap(ptr, n, torch.int32, dev).clone()
                graph.merged_csr = {
                    "node_types": [pl["types"][i] for i in used], "edge_types": slot_ets,
                    "edge_type_ids": {k: int(edge_type_ids[k]) for k in slot_ets},
                    "csr": (take(csr.rowptr, n_dst + 1), take(csr.col, n_e), take(csr.etype, n_e)),
                    "root_type": root_node_type, "root_index": root_index,
                    "root_csr": (take(csr.root_rowptr, b + 1), take(csr.root_col, n_e), take(csr.root_etype, n_e))}
        return graph, root_index, uniq

    def _feature_table(self, node_type: str) -> Optional[torch.Tensor]:
        if not hasattr(self, "_dev_feats"):
            self._dev_feats: Dict[str, torch.Tensor] = {}
        if node_type not in self._dev_feats and self.features.get(node_type) is not None:
            self._dev_feats[node_type] = torch.from_numpy(
                np.ascontiguousarray(self.features[node_type], dtype=np.float32)).to(self.engine.device)
        return self._dev_feats.get(node_type)

    # ---- KHopSamplerService surface ------------------------------------------------------------------------------
    def getKHopSubgraphForRootNodes(self, root_ids: Sequence[int], root_node_type: str,
                                    dag: SamplingOpDAG) -> List[wire.RootedNodeNeighborhood]:
        roots = torch.tensor(np.asarray(root_ids, dtype=np.int64).astype(np.uint32).view(np.int32))
        res = self.run_dag(roots, dag)
        self.engine.synchronize()
        host = {k: (r.frontier.cpu().numpy().view(np.uint32), r.nbr.cpu().numpy().view(np.uint32)) for k, r in res.items()}
        out = []
        for i, root in enumerate(root_ids):
            edges: Set[Tuple[int, int, int]] = set()
            nodes: Set[Tuple[int, int]] = {(int(root), self.node_types[root_node_type])}
            for name, (front, nbr) in host.items():
                op = dag.nodes[name].sampling_op
                cet = self.condensed_edge_types[op.edge_type]
                outgoing = op.sampling_direction == OUTGOING
                got_type = self.node_types[op.edge_type.dst_node_type if outgoing else op.edge_type.src_node_type]
                fr, nb = front[i], nbr[i]
                for q in np.flatnonzero(fr != INVALID):
                    for v in nb[q][nb[q] != INVALID].tolist():
                        edges.add((int(fr[q]), v, cet) if outgoing else (v, int(fr[q]), cet))
                        nodes.add((v, got_type))
            out.append(self._message(int(root), root_node_type, sorted(nodes), sorted(edges)))
        return out

    def encode_records(self, root_ids: Sequence[int], root_node_type: str, dag: SamplingOpDAG,
                       tfrecord_frame: bool = True) -> List[bytes]:
        """the same RootedNodeNeighborhood messages as getKHopSubgraphForRootNodes, serialized ON THE DEVICE — one
        bytes object per root (the host only slices the finished byte string)"""
        out, off = self.encode_records_device(root_ids, root_node_type, dag, tfrecord_frame)
        blob, off = out.cpu().numpy().tobytes(), off.cpu().numpy()
        return [blob[int(off[i]):int(off[i + 1])] for i in range(len(root_ids))]

    def encode_records_device(self, root_ids: Sequence[int], root_node_type: str, dag: SamplingOpDAG,
                              tfrecord_frame: bool = True):
        """-> (uint8 device tensor: the records back to back, int64 device tensor rec_off[b + 1])
        (gigl_typed_records_encode: per root the distinct typed nodes and edges of every op, sorted, with the nodes'
        feature rows, the edges' feature rows joined by binary search in their type's edge list)"""
        roots = torch.tensor(np.asarray(root_ids, dtype=np.int64).astype(np.uint32).view(np.int32))
        res = self.run_dag(roots, dag)
        n_types = max(self.node_types.values()) + 1
        by_cnt = {c: t for t, c in self.node_types.items()}
        feats = [self._feature_table(by_cnt[c]) if c in by_cnt else None for c in range(n_types)]
        ops = []
        for name, r in res.items():
            op = dag.nodes[name].sampling_op
            outgoing = op.sampling_direction == OUTGOING
            got = self.node_types[op.edge_type.dst_node_type if outgoing else op.edge_type.src_node_type]
            ops.append((r.frontier, r.nbr, self.condensed_edge_types[op.edge_type], got, outgoing))
        n_et = max(self.condensed_edge_types.values()) + 1
        by_c = {c: et for et, c in self.condensed_edge_types.items()}
        edge_feats = [self._key(by_c[c], OUTGOING) if c in by_c and self._has_edge_feats.get(by_c[c]) else None
                      for c in range(n_et)]
        return self.engine.encode_typed_records(roots.to(self.engine.device), self.node_types[root_node_type], ops,
                                                feats, tfrecord_frame=tfrecord_frame, edge_feats=edge_feats)

    # ---- typed training samples (GraphDBNodeAnchorBasedLinkPredictionTask.scala:80-116, 333-470;
    #      GraphDBSampler.samplePositiveEdgeNeighborhoods :175-218) ------------------------------------------------
    POSITIVE_OP_NAME = "samplePositiveEdges"

    def _run_nablp(self, root_ids: Sequence[int], positive_edge_type: EdgeType, num_positives: int,
                   root_dag: SamplingOpDAG, positive_dag: SamplingOpDAG):
        """-> (roots int32 [b], root DAG results, positives OpResult ([b, 1] / [b, 1, P]), positives' DAG results over
        the b*P flattened positives).  The positives are sampled OUTGOING along the supervision edge type with the
        sampler's rule (K = 2 * root id, counter 1 + the number of ops of the root's DAG: the op after them); every
        positive's neighbourhood is the one it gets as a root of its own node type's DAG."""
        for name in positive_dag.root_op_names:
            et = positive_dag.nodes[name].sampling_op.edge_type
            if et.dst_node_type != positive_edge_type.dst_node_type:
                raise ValueError(f"SamplingOpDAG should have edgeType {positive_edge_type} matching for all sampling "
                                 f"ops. Mismatch for {name}")
        eng = self.engine
        roots = torch.tensor(np.asarray(root_ids, dtype=np.int64).astype(np.uint32).view(np.int32)).to(eng.device)
        res = self.run_dag(roots, root_dag)
        pos = self.sample_positives(roots, positive_edge_type, num_positives, root_dag)
        pos_res = self.run_dag(pos.nbr.reshape(-1), positive_dag)  # INVALID positives sample nothing
        return roots, res, pos, pos_res

    def sample_positives(self, root_ids, positive_edge_type: EdgeType, num_positives: int, root_dag: SamplingOpDAG) -> OpResult:
        """the positive-edge op of the typed link-prediction samples alone: `num_positives` OUTGOING neighbours of every
        root along the supervision edge type (K = 2 * root id, counter 1 + the number of ops of the root's DAG)"""
        eng = self.engine
        roots = root_ids if isinstance(root_ids, torch.Tensor) else \
            torch.tensor(np.asarray(root_ids, dtype=np.int64).astype(np.uint32).view(np.int32))
        roots = roots.to(device=eng.device, dtype=torch.int32).contiguous()
        b, P = int(roots.numel()), int(num_positives)
        front = roots.view(b, 1).contiguous()
        ksum = (front + roots.view(b, 1)).contiguous()
        counter = 1 + len(root_dag.op_order)
        nbr, cnt = eng.expand_frontier(front.view(-1), ksum.view(-1), P, self.sampling_seed * counter, 1,
                                       label_edges=self._key(positive_edge_type, OUTGOING))
        return OpResult(front, nbr.view(b, 1, P), cnt.view(b, 1))

    def getNablpSamplesForRootNodes(self, root_ids: Sequence[int], positive_edge_type: EdgeType, num_positives: int,
                                    root_dag: SamplingOpDAG, positive_dag: SamplingOpDAG
                                    ) -> List[wire.NodeAnchorBasedLinkPredictionSample]:
        """host assembly: root node, pos_edges (root -> positive, hydrated), neighbourhood = the root's merged with its
        positives' (mergeGraphs); a root without positives gets empty pos_edges and its own neighbourhood"""
        root_type = positive_edge_type.src_node_type
        roots, res, pos, pos_res = self._run_nablp(root_ids, positive_edge_type, num_positives, root_dag, positive_dag)
        self.engine.synchronize()
        b, P = len(root_ids), int(num_positives)
        to_h = lambda t: t.cpu().numpy().view(np.uint32)
        pos_h = to_h(pos.nbr).reshape(b, P)
        cet_pos = self.condensed_edge_types[positive_edge_type]
        out = []
        parts = [(root_dag, {k: (to_h(r.frontier), to_h(r.nbr)) for k, r in res.items()}, 1),
                 (positive_dag, {k: (to_h(r.frontier), to_h(r.nbr)) for k, r in pos_res.items()}, P)]
        for i, root in enumerate(root_ids):
            edges: Set[Tuple[int, int, int]] = set()
            nodes: Set[Tuple[int, int]] = {(int(root), self.node_types[root_type])}
            pos_edges = sorted({(int(root), int(v), cet_pos) for v in pos_h[i] if v != INVALID})
            nodes |= {(d, self.node_types[positive_edge_type.dst_node_type]) for _, d, _ in pos_edges}
            for dag, host, rows in parts:
                for name, (front, nbr) in host.items():
                    op = dag.nodes[name].sampling_op
                    cet = self.condensed_edge_types[op.edge_type]
                    outgoing = op.sampling_direction == OUTGOING
                    got_type = self.node_types[op.edge_type.dst_node_type if outgoing else op.edge_type.src_node_type]
                    for row in range(i * rows, (i + 1) * rows):
                        fr, nb = front[row], nbr[row]
                        for q in np.flatnonzero(fr != INVALID):
                            for v in nb[q][nb[q] != INVALID].tolist():
                                edges.add((int(fr[q]), v, cet) if outgoing else (v, int(fr[q]), cet))
                                nodes.add((v, got_type))
            rnn = self._message(int(root), root_type, sorted(nodes), sorted(edges))
            out.append(wire.NodeAnchorBasedLinkPredictionSample(
                root_node=rnn.root_node, neighborhood=rnn.neighborhood,
                pos_edges=[wire.Edge(src_node_id=s_, dst_node_id=d_, condensed_edge_type=c_,
                                     feature_values=self._edge_rows.get((s_, d_, c_), wire._EMPTY_F32))
                           for s_, d_, c_ in pos_edges]))
        return out

    def encode_nablp_records(self, root_ids: Sequence[int], positive_edge_type: EdgeType, num_positives: int,
                             root_dag: SamplingOpDAG, positive_dag: SamplingOpDAG, tfrecord_frame: bool = True):
        """the messages of getNablpSamplesForRootNodes serialized ON THE DEVICE -> (list of record bytes, number of
        positives per root)"""
        out, off, n_pos = self.encode_nablp_records_device(root_ids, positive_edge_type, num_positives, root_dag,
                                                           positive_dag, tfrecord_frame)
        blob, off = out.cpu().numpy().tobytes(), off.cpu().numpy()
        return [blob[int(off[i]):int(off[i + 1])] for i in range(len(root_ids))], n_pos.cpu().numpy()

    def encode_nablp_records_device(self, root_ids: Sequence[int], positive_edge_type: EdgeType, num_positives: int,
                                    root_dag: SamplingOpDAG, positive_dag: SamplingOpDAG, tfrecord_frame: bool = True):
        """gigl_typed_samples_encode, kind NODE_ANCHOR_LINK_PRED -> (uint8 device records, int64 device rec_off[b + 1],
        int64 device positives per root)"""
        from ._lib import REC_NODE_ANCHOR_LINK_PRED
        root_type = positive_edge_type.src_node_type
        roots, res, pos, pos_res = self._run_nablp(root_ids, positive_edge_type, num_positives, root_dag, positive_dag)
        b, P = len(root_ids), int(num_positives)
        n_types = max(self.node_types.values()) + 1
        by_cnt = {c: t for t, c in self.node_types.items()}
        feats = [self._feature_table(by_cnt[c]) if c in by_cnt else None for c in range(n_types)]

        def typed(dag, name, r, rows):
            op = dag.nodes[name].sampling_op
            outgoing = op.sampling_direction == OUTGOING
            got = self.node_types[op.edge_type.dst_node_type if outgoing else op.edge_type.src_node_type]
            f = int(r.nbr.shape[-1])
            # (the P positives of a root are consecutive rows: side by side they are one [b, P*w] frontier)
            return (r.frontier.reshape(b, -1), r.nbr.reshape(b, -1, f), self.condensed_edge_types[op.edge_type], got,
                    outgoing)
        ops = [typed(root_dag, n_, r, 1) for n_, r in res.items()]
        ops.append((pos.frontier, pos.nbr, self.condensed_edge_types[positive_edge_type],
                    self.node_types[positive_edge_type.dst_node_type], True, True))
        ops += [typed(positive_dag, n_, r, P) for n_, r in pos_res.items()]
        n_et = max(self.condensed_edge_types.values()) + 1
        by_c = {c: et for et, c in self.condensed_edge_types.items()}
        edge_feats = [self._key(by_c[c], OUTGOING) if c in by_c and self._has_edge_feats.get(by_c[c]) else None
                      for c in range(n_et)]
        out, off = self.engine.encode_typed_records(roots, self.node_types[root_type], ops, feats,
                                                    tfrecord_frame=tfrecord_frame, edge_feats=edge_feats,
                                                    kind=REC_NODE_ANCHOR_LINK_PRED)
        n_pos = (pos.nbr.view(b, P) != -1).sum(dim=1)
        return out, off, n_pos

    def write_tfrecords(self, path: str, root_ids: Sequence[int], root_node_type: str, dag: SamplingOpDAG) -> int:
        """a part file of the sampler job: the device-encoded frames of encode_records written back to back"""
        frames = self.encode_records(root_ids, root_node_type, dag, tfrecord_frame=True)
        with open(path, "wb") as f:
            for fr in frames:
                f.write(fr)
        return len(frames)

    def getKHopSubgraphForRootNode(self, root_id: int, root_node_type: str, dag: SamplingOpDAG):
        return self.getKHopSubgraphForRootNodes([root_id], root_node_type, dag)[0]

    def _message(self, root: int, root_type: str, nodes, edges) -> wire.RootedNodeNeighborhood:
        by_cnt = {c: t for t, c in self.node_types.items()}

        def node(v: int, cnt: int) -> wire.Node:
            x = self.features.get(by_cnt[cnt])
            fv = x[v] if x is not None else wire._EMPTY_F32
            return wire.Node(node_id=v, condensed_node_type=cnt, feature_values=np.asarray(fv, dtype=np.float32))
        return wire.RootedNodeNeighborhood(
            root_node=node(root, self.node_types[root_type]),
            neighborhood=wire.Graph(nodes=[node(v, c) for v, c in nodes],
                                    edges=[wire.Edge(src_node_id=s, dst_node_id=d, condensed_edge_type=c,
                                                     feature_values=self._edge_rows.get((s, d, c), wire._EMPTY_F32))
                                           for s, d, c in edges]))
