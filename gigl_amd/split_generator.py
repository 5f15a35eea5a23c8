"""SplitGenerator — drop-in for the reference component between the sampler and the trainer

    python -m gigl.src.split_generator.split_generator --job_name --task_config_uri --resource_config_uri
    Scala job: scala/split_generator/src/main/scala/Main.scala:14-38, lib/SplitGeneratorTaskRunner.scala:20-94

Restated reference pieces (paths under scala/split_generator/src/main/scala/lib/):
  assigners/AbstractAssigners.scala:30-111            HashingAssigner: slot = floorMod(MurmurHash3.bytesHash(coder(obj)),
                                                      10000), bucket = the span of the cumulative normalised weights
                                                      (float32 arithmetic, math.round) the slot falls in
  assigners/NodeToDatasetSplitHashingAssigner.scala    coder = "<nodeId>-<condensedNodeType>"   (GraphPbWrappers.scala:37-39)
  assigners/TransductiveEdgeToLinkSplitHashingAssigner.scala
                                                      coder = "<src>-<condensedEdgeType>-<dst>" of the canonically
                                                      ordered edge when should_split_edges_symmetrically (:66-78);
                                                      disjoint_train_ratio > 0 adds TRAIN/MESSAGE + TRAIN/SUPERVISION
  split_strategies/TransductiveSupervisedNodeClassificationSplitStrategy.scala:25-49
  split_strategies/InductiveSupervisedNodeClassificationSplitStrategy.scala:20-89
  split_strategies/TransductiveNodeAnchorBasedLinkPredictionSplitStrategy.scala:37-268
  tasks/SupervisedNodeClassificationTask.scala, tasks/NodeAnchorBasedLinkPredictionTask.scala   (input / output URIs)

Parity: the reference's tests for this component are property tests (determinism, symmetric assignment, split
ratios, message-passing visibility rules — scala/split_generator/src/test/scala/*.scala); no reference artefact pins
a concrete hash slot -> "parity unpinned" for the exact assignment.  MurmurHash3.bytesHash is scala.util.hashing's
MurmurHash3_x86_32 with seed 0x3c074a61 (its `arraySeed`), restated here and checked against the public
MurmurHash3_x86_32 vectors (tests/test_split_generator.py).  `subsample` draws scala.util.Random.nextFloat in the
reference (unseeded): any ratio < 1 is therefore non-reproducible there; a seeded numpy generator is used here.

Hashing in bulk: SplitGenerator.run hashes every node / edge of the input samples in one device pass
(gigl_split_hash_slots, csrc/split.hip — the strategies' per-object `assign` calls then hit a cache); the device
kernel, this module's host routine and the C restatement in oracle/ are checked against each other and against the
public MurmurHash3_x86_32 vectors (tests/test_split_generator.py, tests/test_gpu_split.py).
"""
from __future__ import annotations

import argparse
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import wire
from .config import GbmlConfigPbWrapper, _get, resolve_uri, tfrecord_files

TRAIN, VAL, TEST = "train", "val", "test"
MESSAGE, SUPERVISION, MESSAGE_AND_SUPERVISION = "message", "supervision", "message_and_supervision"
HASH_SPACE_GRANULARITY = 10000
SCALA_ARRAY_SEED = 0x3C074A61


def murmur3_bytes_hash(data: bytes, seed: int = SCALA_ARRAY_SEED) -> int:
    """scala.util.hashing.MurmurHash3.bytesHash == MurmurHash3_x86_32; returns a signed int32"""
    c1, c2, m32 = 0xCC9E2D51, 0x1B873593, 0xFFFFFFFF
    h = seed & m32
    n = len(data)
    for i in range(0, n - n % 4, 4):
        k = data[i] | (data[i + 1] << 8) | (data[i + 2] << 16) | (data[i + 3] << 24)
        k = (k * c1) & m32
        k = ((k << 15) | (k >> 17)) & m32
        k = (k * c2) & m32
        h ^= k
        h = ((h << 13) | (h >> 19)) & m32
        h = (h * 5 + 0xE6546B64) & m32
    k = 0
    t = n % 4
    if t:
        base = n - t
        if t == 3:
            k ^= data[base + 2] << 16
        if t >= 2:
            k ^= data[base + 1] << 8
        k ^= data[base]
        k = (k * c1) & m32
        k = ((k << 15) | (k >> 17)) & m32
        k = (k * c2) & m32
        h ^= k
    h ^= n
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & m32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & m32
    h ^= h >> 16
    return h - (1 << 32) if h & 0x80000000 else h


class HashingAssigner:
    """AbstractAssigners.HashingAssigner: buckets in insertion order of bucketWeights (an immutable Map of <= 4
    entries keeps insertion order), transition indices from float32 cumulative normalised weights"""

    def __init__(self, bucket_weights: Sequence[Tuple[object, float]]):
        self.buckets = [b for b, _ in bucket_weights]
        w = np.array([x for _, x in bucket_weights], dtype=np.float32)
        total = np.float32(0)
        for x in w:  # Seq[Float].sum: left fold in float32
            total = np.float32(total + x)
        normed = (w / total).astype(np.float32)
        cum = [np.float32(0)]
        for x in normed:
            cum.append(np.float32(cum[-1] + x))
        # math.round(Float): floor(x + 0.5) as int
        self.indices = [int(np.floor(np.float32(c * np.float32(HASH_SPACE_GRANULARITY)) + np.float32(0.5))) for c in cum]
        self._cache: Dict[bytes, object] = {}

    def _bucket_of_slot(self, slot: int):
        for b, lo, hi in zip(self.buckets, self.indices, self.indices[1:]):
            if lo <= slot < hi:
                return b
        raise AssertionError(f"hash slot {slot} falls in no bucket (indices {self.indices})")

    def prefill(self, engine, a, b=None, condensed_type: int = 0, symmetric: bool = False) -> int:
        """hash many objects at once on the device (gigl_split_hash_slots, csrc/split.hip) and remember their buckets:
        nodes `a` (keys "<id>-<type>") or edges `a` -> `b` (keys "<src>-<type>-<dst>").  The per-object `assign`
        calls of the strategies then hit the cache instead of hashing in Python.  -> objects hashed"""
        import ctypes as C
        import torch
        from ._lib import check
        a = np.ascontiguousarray(a, dtype=np.uint32)
        if a.size == 0:
            return 0
        dev = engine.device
        ta = torch.from_numpy(a.view(np.int32)).to(dev)
        tb = None
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.uint32)
            tb = torch.from_numpy(b.view(np.int32)).to(dev)
        slots = torch.empty(a.size, dtype=torch.int32, device=dev)
        engine._stream.synchronize()
        check(engine._lib.gigl_split_hash_slots(engine._ctx, C.c_void_p(ta.data_ptr()),
                                                C.c_void_p(tb.data_ptr()) if tb is not None else None, a.size,
                                                int(condensed_type), 1 if symmetric else 0,
                                                C.c_void_p(slots.data_ptr())), engine._ctx)
        engine._stream.synchronize()
        sl = slots.cpu().numpy()
        bounds = np.asarray(self.indices)
        which = np.searchsorted(bounds, sl, side="right") - 1
        if b is None:
            for x, w in zip(a.tolist(), which.tolist()):
                self._cache[node_unique_id(x, condensed_type)] = self.buckets[w]
        else:
            for x, y, w in zip(a.tolist(), b.tolist(), which.tolist()):
                if symmetric and x > y:
                    x, y = y, x
                self._cache[edge_unique_id(x, y, condensed_type)] = self.buckets[w]
        return int(a.size)

    def assign_bytes(self, byte_string: bytes):
        hit = self._cache.get(byte_string)
        if hit is not None:
            return hit
        slot = murmur3_bytes_hash(byte_string) % HASH_SPACE_GRANULARITY  # Python % == floorMod for a positive modulus
        for b, lo, hi in zip(self.buckets, self.indices, self.indices[1:]):
            if lo <= slot < hi:
                self._cache[byte_string] = b
                return b
        raise AssertionError(f"hash slot {slot} falls in no bucket (indices {self.indices})")


def node_unique_id(node_id: int, condensed_node_type: Optional[int]) -> bytes:
    return f"{node_id}-{condensed_node_type or 0}".encode()


def edge_unique_id(src: int, dst: int, condensed_edge_type: Optional[int]) -> bytes:
    return f"{src}-{condensed_edge_type or 0}-{dst}".encode()


class NodeToDatasetSplitHashingAssigner(HashingAssigner):
    def __init__(self, assigner_args: Dict[str, str]):
        f = lambda k, d: float(np.float32(assigner_args.get(k, d)))
        super().__init__([(TRAIN, f("train_split", "0.8")), (VAL, f("val_split", "0.1")), (TEST, f("test_split", "0.1"))])

    def assign(self, node: wire.Node) -> str:
        return self.assign_bytes(node_unique_id(node.node_id, node.condensed_node_type))

    def assign_id(self, node_id: int, condensed_node_type: Optional[int] = 0) -> str:
        return self.assign_bytes(node_unique_id(node_id, condensed_node_type))


class TransductiveEdgeToLinkSplitHashingAssigner(HashingAssigner):
    """-> (dataset split, link usage)"""

    def __init__(self, assigner_args: Dict[str, str]):
        f32 = np.float32
        train = f32(assigner_args.get("train_split", "0.8"))
        val = f32(assigner_args.get("val_split", "0.1"))
        test = f32(assigner_args.get("test_split", "0.1"))
        disjoint = f32(assigner_args.get("disjoint_train_ratio", "0.0"))
        self.symmetric = str(assigner_args.get("should_split_edges_symmetrically", "True")).lower() == "true"
        if disjoint > 0:
            sup = f32(disjoint * train)
            msg = f32(train - sup)
            weights = [((TRAIN, MESSAGE), float(msg)), ((TRAIN, SUPERVISION), float(sup)),
                       ((VAL, MESSAGE_AND_SUPERVISION), float(val)), ((TEST, MESSAGE_AND_SUPERVISION), float(test))]
        else:
            weights = [((TRAIN, MESSAGE_AND_SUPERVISION), float(train)), ((VAL, MESSAGE_AND_SUPERVISION), float(val)),
                       ((TEST, MESSAGE_AND_SUPERVISION), float(test))]
        super().__init__(weights)

    def assign(self, edge: wire.Edge) -> Tuple[str, str]:
        s, d = edge.src_node_id, edge.dst_node_id
        if self.symmetric and not s <= d:  # hash the canonically ordered edge: a->b and b->a land together
            s, d = d, s
        return self.assign_bytes(edge_unique_id(s, d, edge.condensed_edge_type))


class UserDefinedLabelsEdgeToLinkSplitHashingAssigner(HashingAssigner):
    """user-provided pos / hard-neg edges -> (TRAIN | VAL | TEST, SUPERVISION)
    (lib/assigners/UserDefinedLabelsEdgeToLinkSplitHashingAssigner.scala:16-60; symmetric hashing defaults to False)"""

    def __init__(self, assigner_args: Dict[str, str]):
        f32 = np.float32
        weights = [((TRAIN, SUPERVISION), float(f32(assigner_args.get("train_split", "0.8")))),
                   ((VAL, SUPERVISION), float(f32(assigner_args.get("val_split", "0.1")))),
                   ((TEST, SUPERVISION), float(f32(assigner_args.get("test_split", "0.1"))))]
        self.symmetric = str(assigner_args.get("should_split_edges_symmetrically", "False")).lower() == "true"
        super().__init__(weights)

    def assign(self, edge: wire.Edge) -> Tuple[str, str]:
        s, d = edge.src_node_id, edge.dst_node_id
        if self.symmetric and not s <= d:
            s, d = d, s
        return self.assign_bytes(edge_unique_id(s, d, edge.condensed_edge_type))


class _Subsampler:
    def __init__(self, args: Dict[str, str], prefix: str = "", seed: int = 42):
        self.ratio = {TRAIN: float(args.get(prefix + "train_subsampling_ratio", "1.0")),
                      VAL: float(args.get(prefix + "val_subsampling_ratio", "1.0")),
                      TEST: float(args.get(prefix + "test_subsampling_ratio", "1.0"))}
        self.rng = np.random.default_rng(seed)

    def __call__(self, samples: list, split: str) -> list:
        """SplitStrategy.subsample (SplitStrategy.scala:48-64) — note the reference DROPS the sample when the draw is
        BELOW the ratio (ratio < 1), kept verbatim"""
        r = self.ratio[split]
        if r < 1.0 and float(self.rng.random()) < r:
            return []
        return samples


class TransductiveSupervisedNodeClassificationSplitStrategy:
    def __init__(self, args: Dict[str, str], assigner: NodeToDatasetSplitHashingAssigner):
        self.assigner, self.subsample = assigner, _Subsampler(args)

    def split_training_sample(self, sample: wire.SupervisedNodeClassificationSample, split: str):
        if sample.root_node is None:
            raise RuntimeError("Root node does not exist for sample.")
        out = [sample] if self.assigner.assign(sample.root_node) == split else []
        return self.subsample(out, split)


class InductiveSupervisedNodeClassificationSplitStrategy(TransductiveSupervisedNodeClassificationSplitStrategy):
    def split_training_sample(self, sample: wire.SupervisedNodeClassificationSample, split: str):
        if sample.root_node is None:
            raise RuntimeError("Root node does not exist for sample.")
        out = []
        if self.assigner.assign(sample.root_node) == split:
            if sample.neighborhood is None:
                raise RuntimeError("Neighborhood does not exist in the sample")
            nodes = [n for n in sample.neighborhood.nodes if self.assigner.assign(n) == split]
            # homogeneous graphs: both endpoints carry the (single) condensed node type 0
            edges = [e for e in sample.neighborhood.edges
                     if self.assigner.assign_id(e.src_node_id, 0) == split and self.assigner.assign_id(e.dst_node_id, 0) == split]
            out = [wire.SupervisedNodeClassificationSample(root_node=sample.root_node, root_node_labels=sample.root_node_labels,
                                                           neighborhood=wire.Graph(nodes=nodes, edges=edges))]
        return self.subsample(out, split)


class TransductiveNodeAnchorBasedLinkPredictionSplitStrategy:
    def __init__(self, args: Dict[str, str], assigner: TransductiveEdgeToLinkSplitHashingAssigner):
        self.assigner = assigner
        self.is_disjoint_mode = str(args.get("is_disjoint_mode", "false")).lower() == "true"
        self.subsample = _Subsampler(args)
        self.subsample_rn = _Subsampler(args, prefix="random_negative_")

    def _message_edge_visible(self, edge: wire.Edge, split: str) -> bool:
        ds, usage = self.assigner.assign(edge)
        if split == TRAIN:
            return ds == TRAIN and (not self.is_disjoint_mode or usage == MESSAGE)
        if split == VAL:
            return ds == TRAIN
        return ds in (TRAIN, VAL)  # test edges are never used for message passing

    def _supervision_edge_in_split(self, edge: wire.Edge, split: str) -> bool:
        ds, usage = self.assigner.assign(edge)
        is_message_edge = usage == MESSAGE and split == TRAIN  # training MESSAGE edges are not supervision
        return not is_message_edge and ds == split

    @staticmethod
    def _graph(sample_graph: Optional[wire.Graph], root: Optional[wire.Node], mp_edges, sup_edges) -> wire.Graph:
        if sample_graph is None:
            raise RuntimeError("Neighborhood does not exist in the sample")
        if root is None:
            raise RuntimeError("Root Node does not exist for the sample")
        # featureless Node(nodeId, condensedNodeType) set: the root as it is, edge endpoints with the condensed node
        # type graph metadata gives the edge type's endpoints (getFeaturelessNodePbsFromEdge; homogeneous: Some(0))
        keep = {(root.node_id, root.condensed_node_type)}
        for e in list(mp_edges) + list(sup_edges):
            keep.add((e.src_node_id, 0))
            keep.add((e.dst_node_id, 0))
        nodes = [n for n in sample_graph.nodes if (n.node_id, n.condensed_node_type) in keep]
        return wire.Graph(nodes=nodes, edges=list(mp_edges))

    def split_training_sample(self, sample: wire.NodeAnchorBasedLinkPredictionSample, split: str):
        pos = [e for e in sample.pos_edges if self._supervision_edge_in_split(e, split)]
        neg = [e for e in sample.neg_edges if self._supervision_edge_in_split(e, split)]
        hard = [e for e in sample.hard_neg_edges if self._supervision_edge_in_split(e, split)]
        if not pos and split == TRAIN:  # a train sample needs a positive for the loss
            return []
        if sample.neighborhood is None:
            raise RuntimeError("Neighborhood does not exist in the sample")
        mp = [e for e in sample.neighborhood.edges if self._message_edge_visible(e, split)]
        out = [wire.NodeAnchorBasedLinkPredictionSample(
            root_node=sample.root_node, hard_neg_edges=hard, pos_edges=pos, neg_edges=neg,
            neighborhood=self._graph(sample.neighborhood, sample.root_node, mp, pos + neg + hard))]
        return self.subsample(out, split)

    def split_rooted_node_neighborhood_training_sample(self, sample: wire.RootedNodeNeighborhood, split: str):
        if sample.neighborhood is None:
            raise RuntimeError("Neighborhood does not exist in the sample")
        mp = [e for e in sample.neighborhood.edges if self._message_edge_visible(e, split)]
        out = [wire.RootedNodeNeighborhood(root_node=sample.root_node,
                                           neighborhood=self._graph(sample.neighborhood, sample.root_node, mp, []))]
        return self.subsample_rn(out, split)


class UserDefinedLabelsNodeAnchorBasedLinkPredictionSplitStrategy(TransductiveNodeAnchorBasedLinkPredictionSplitStrategy):
    """samples whose pos_edges / hard_neg_edges are user-defined labels
    (lib/split_strategies/UserDefinedLabelsNodeAnchorBasedLinkPredictionSplitStrategy.scala:21-153): the neighbourhood
    is message-passing structure and stays whole in every split; only the label edges are assigned to splits; a train
    sample without a positive is dropped, val / test samples may come out without any label edge"""

    def _label_edge_in_split(self, edge: wire.Edge, split: str) -> bool:
        ds, usage = self.assigner.assign(edge)
        if usage != SUPERVISION:
            raise RuntimeError(f"Unexpected edgeUsage type {usage} for supervision edge {edge}")
        return ds == split

    def split_training_sample(self, sample: wire.NodeAnchorBasedLinkPredictionSample, split: str):
        pos = [e for e in sample.pos_edges if self._label_edge_in_split(e, split)]
        neg = [e for e in sample.neg_edges if self._label_edge_in_split(e, split)]
        hard = [e for e in sample.hard_neg_edges if self._label_edge_in_split(e, split)]
        if not pos and split == TRAIN:
            return []
        out = [wire.NodeAnchorBasedLinkPredictionSample(root_node=sample.root_node, hard_neg_edges=hard, pos_edges=pos,
                                                        neg_edges=neg, neighborhood=sample.neighborhood)]
        return self.subsample(out, split)

    def split_rooted_node_neighborhood_training_sample(self, sample: wire.RootedNodeNeighborhood, split: str):
        out = [wire.RootedNodeNeighborhood(root_node=sample.root_node, neighborhood=sample.neighborhood)]
        return self.subsample_rn(out, split)


class UDLAnchorBasedSupervisionEdgeSplitStrategy:
    """user-defined labels split by ANCHOR node (lib/split_strategies/UDLAnchorBasedSupervisionEdgeSplitStrategy.scala:
    28-233): a sample goes, whole, to the split its root node hashes to (NodeToDatasetSplitHashingAssigner); in the
    splits named by should_filter_{train,val,test} (default all) the neighbourhood loses every edge that is also a
    pos / neg / hard-neg label edge (and its reverse unless should_filter_reverse_supervision_edge=false) together with
    the nodes no remaining edge touches; a sample whose neighbourhood has no edge left — or, in train, no positive —
    is dropped.  Rooted neighbourhoods pass through unmasked."""

    def __init__(self, args: Dict[str, str], assigner: NodeToDatasetSplitHashingAssigner):
        self.assigner = assigner
        flag = lambda k, d="true": str(args.get(k, d)).lower() == "true"
        self.filter_reverse = flag("should_filter_reverse_supervision_edge")
        self.filter_in = {TRAIN: flag("should_filter_train"), VAL: flag("should_filter_val"),
                          TEST: flag("should_filter_test")}
        self.subsample = _Subsampler(args)
        self.subsample_rn = _Subsampler(args, prefix="random_negative_")

    def _without_label_overlap(self, sample: wire.NodeAnchorBasedLinkPredictionSample):
        labels = list(sample.pos_edges) + list(sample.neg_edges) + list(sample.hard_neg_edges)
        unusable = {(e.src_node_id, e.dst_node_id) for e in labels}
        if self.filter_reverse:
            unusable |= {(e.dst_node_id, e.src_node_id) for e in labels}
        edges = [e for e in sample.neighborhood.edges if (e.src_node_id, e.dst_node_id) not in unusable]
        touched = {v for e in edges for v in (e.src_node_id, e.dst_node_id)}
        nodes = [n for n in sample.neighborhood.nodes if n.node_id in touched]
        return wire.NodeAnchorBasedLinkPredictionSample(
            root_node=sample.root_node, hard_neg_edges=sample.hard_neg_edges, pos_edges=sample.pos_edges,
            neg_edges=sample.neg_edges, neighborhood=wire.Graph(nodes=nodes, edges=edges))

    def split_training_sample(self, sample: wire.NodeAnchorBasedLinkPredictionSample, split: str):
        if sample.root_node is None:
            raise RuntimeError("Root node does not exist for sample.")
        if sample.neighborhood is None:
            raise RuntimeError("Neighborhood does not exist in the sample")
        if self.assigner.assign(sample.root_node) != split:
            return []
        usable = self._without_label_overlap(sample) if self.filter_in[split] else sample
        if not usable.neighborhood.edges or (not usable.pos_edges and split == TRAIN):
            return []
        return self.subsample([usable], split)

    def split_rooted_node_neighborhood_training_sample(self, sample: wire.RootedNodeNeighborhood, split: str):
        return self.subsample_rn([sample], split)


_ASSIGNERS = {"NodeToDatasetSplitHashingAssigner": NodeToDatasetSplitHashingAssigner,
              "TransductiveEdgeToLinkSplitHashingAssigner": TransductiveEdgeToLinkSplitHashingAssigner,
              "UserDefinedLabelsEdgeToLinkSplitHashingAssigner": UserDefinedLabelsEdgeToLinkSplitHashingAssigner}
_STRATEGIES = {"TransductiveSupervisedNodeClassificationSplitStrategy": TransductiveSupervisedNodeClassificationSplitStrategy,
               "InductiveSupervisedNodeClassificationSplitStrategy": InductiveSupervisedNodeClassificationSplitStrategy,
               "TransductiveNodeAnchorBasedLinkPredictionSplitStrategy": TransductiveNodeAnchorBasedLinkPredictionSplitStrategy,
               "UserDefinedLabelsNodeAnchorBasedLinkPredictionSplitStrategy":
                   UserDefinedLabelsNodeAnchorBasedLinkPredictionSplitStrategy,
               "UDLAnchorBasedSupervisionEdgeSplitStrategy": UDLAnchorBasedSupervisionEdgeSplitStrategy}


def build_strategy(cfg: GbmlConfigPbWrapper):
    """SplitGeneratorTaskRunner (lib/SplitGeneratorTaskRunner.scala:20-94): class paths are the reference's Scala
    names (splitgenerator.lib.assigners.* / .split_strategies.*); the last path component selects the class"""
    sg = _get(cfg.doc, "datasetConfig.splitGeneratorConfig", {}) or {}
    a_name = str(sg.get("assignerClsPath", "")).rsplit(".", 1)[-1]
    s_name = str(sg.get("splitStrategyClsPath", "")).rsplit(".", 1)[-1]
    if a_name not in _ASSIGNERS or s_name not in _STRATEGIES:
        raise NotImplementedError(f"split generator classes {a_name!r} / {s_name!r} are not implemented "
                                  f"(have {sorted(_ASSIGNERS)} / {sorted(_STRATEGIES)})")
    assigner = _ASSIGNERS[a_name]({k: str(v) for k, v in (sg.get("assignerArgs") or {}).items()})
    return _STRATEGIES[s_name]({k: str(v) for k, v in (sg.get("splitStrategyArgs") or {}).items()}, assigner)


def _write_dir(prefix: str, payloads: List[bytes]) -> List[str]:
    is_dir = prefix.endswith("/") or prefix.endswith(os.sep)
    d = prefix if is_dir else os.path.dirname(prefix)
    os.makedirs(d or ".", exist_ok=True)
    for old in tfrecord_files(prefix):
        os.remove(old)
    name = os.path.join(prefix, "part-00000.tfrecord") if is_dir else f"{prefix}00000.tfrecord"
    wire.write_tfrecords(name, payloads)
    return [name]


def _prefill_assigner(strat, samples) -> int:
    """hash every node / edge the strategy will ask about in ONE device pass (HashingAssigner.prefill); without a HIP
    device the assigner hashes on demand on the host, as the reference's JVM does"""
    assigner = getattr(strat, "assigner", None)
    try:
        import torch
        if assigner is None or not torch.cuda.is_available():
            return 0
        from .engine import default_engine
        eng = default_engine(torch.device("cuda", torch.cuda.current_device()))
    except Exception:  # noqa: BLE001 — no device / no library: host hashing
        return 0
    try:
        return _prefill(assigner, eng, samples)
    except Exception:  # noqa: BLE001 — any device-side failure: the assigner hashes on demand on the host instead
        return 0


def _prefill(assigner, eng, samples) -> int:
    graphs = [g for g in (getattr(smp, "neighborhood", None) for smp in samples) if g is not None]
    if isinstance(assigner, NodeToDatasetSplitHashingAssigner):
        # ids come from Node records only: they carry their condensed type (an edge's endpoints are nodes of the same
        # neighbourhood, so they are all there — TaskOutputValidator.scala:84-107 — under their real type)
        by_type: Dict[int, set] = {}
        for smp in samples:
            r = getattr(smp, "root_node", None)
            if r is not None:
                by_type.setdefault(r.condensed_node_type or 0, set()).add(r.node_id)
        for g in graphs:
            for nd in g.nodes:
                by_type.setdefault(nd.condensed_node_type or 0, set()).add(nd.node_id)
        return sum(assigner.prefill(eng, np.fromiter(ids, dtype=np.uint32, count=len(ids)), condensed_type=t)
                   for t, ids in by_type.items())
    by_type_e: Dict[int, set] = {}
    for smp in samples:
        for fld in ("pos_edges", "hard_neg_edges", "neg_edges"):
            for e in getattr(smp, fld, None) or ():
                by_type_e.setdefault(e.condensed_edge_type or 0, set()).add((e.src_node_id, e.dst_node_id))
    for g in graphs:
        for e in g.edges:
            by_type_e.setdefault(e.condensed_edge_type or 0, set()).add((e.src_node_id, e.dst_node_id))
    n = 0
    for t, pairs in by_type_e.items():
        arr = np.array(sorted(pairs), dtype=np.uint32).reshape(-1, 2)
        n += assigner.prefill(eng, arr[:, 0], arr[:, 1], condensed_type=t, symmetric=getattr(assigner, "symmetric", False))
    return n


def _split_and_write(in_prefix: str, out_prefixes: Dict[str, str], decode: Callable, split_fn: Callable,
                     strat=None):
    """SplitGeneratorTask.splitSamplesAndWriteToOutputPath (lib/tasks/SplitGeneratorTask.scala:60-103)"""
    samples = [decode(r) for f in tfrecord_files(in_prefix) for r in wire.read_tfrecords(f)]
    if strat is not None:
        _prefill_assigner(strat, samples)
    files = {}
    for split in (TRAIN, VAL, TEST):
        out = [s.SerializeToString() for smp in samples for s in split_fn(smp, split)]
        files[split] = _write_dir(out_prefixes[split], out)
    return files


class SplitGenerator:
    def run(self, applied_task_identifier: str, task_config_uri: str, resource_config_uri: Optional[str] = None, *,
            uri_base: Optional[str] = None) -> Dict[str, Dict[str, List[str]]]:
        cfg = GbmlConfigPbWrapper.from_uri(task_config_uri, uri_base=uri_base)
        strat = build_strategy(cfg)
        res = lambda u: resolve_uri(u, cfg.uri_base)
        dm = _get(cfg.doc, "sharedConfig.datasetMetadata", {}) or {}
        if cfg.task_kind == "node_classification":
            ds = dm["supervisedNodeClassificationDataset"]
            outs = {TRAIN: res(ds["trainDataUri"]), VAL: res(ds["valDataUri"]), TEST: res(ds["testDataUri"])}
            return {"main": _split_and_write(cfg.labeled_tfrecord_uri_prefix, outs,
                                             wire.SupervisedNodeClassificationSample.FromString,
                                             strat.split_training_sample, strat)}
        ds = dm["nodeAnchorBasedLinkPredictionDataset"]
        outs = {TRAIN: res(ds["trainMainDataUri"]), VAL: res(ds["valMainDataUri"]), TEST: res(ds["testMainDataUri"])}
        files = {"main": _split_and_write(cfg.nablp_tfrecord_uri_prefix, outs,
                                          wire.NodeAnchorBasedLinkPredictionSample.FromString,
                                          strat.split_training_sample, strat)}
        for node_type, in_prefix in cfg.random_negative_tfrecord_uri_prefixes.items():
            outs = {TRAIN: res(ds["trainNodeTypeToRandomNegativeDataUri"][node_type]),
                    VAL: res(ds["valNodeTypeToRandomNegativeDataUri"][node_type]),
                    TEST: res(ds["testNodeTypeToRandomNegativeDataUri"][node_type])}
            files[f"random_negative/{node_type}"] = _split_and_write(
                in_prefix, outs, wire.RootedNodeNeighborhood.FromString,
                strat.split_rooted_node_neighborhood_training_sample, strat)
        return files


def main(argv=None):
    ap = argparse.ArgumentParser(description="split generator (drop-in for gigl.src.split_generator)")
    ap.add_argument("--job_name", required=True)
    ap.add_argument("--task_config_uri", required=True)
    ap.add_argument("--resource_config_uri", default=None)
    ap.add_argument("--uri_base", default=None)
    a = ap.parse_args(argv)
    for k, v in SplitGenerator().run(a.job_name, a.task_config_uri, a.resource_config_uri, uri_base=a.uri_base).items():
        print(k, {s: len(f) for s, f in v.items()})


if __name__ == "__main__":
    main()
