"""The in-HBM dataflow behind the drop-in entry points (Inferencer / Trainer): the job's graph and feature table are
read ONCE from the Data Preprocessor's tables into HBM, and every batch is sampled, union-ed and pushed through the
model there — the path `bench.py` measures (gigl_sage_plan_run / gigl_gat_plan_* / gigl_dist_plan_run) — instead of
travelling device -> TFRecord part files -> host parse / collate -> device.

Reference dataflow this stands in for (paths relative to the reference root):
  SubgraphSampler (Scala/Spark) writes RootedNodeNeighborhood TFRecords for every node
      scala/subgraph_sampler/.../SGSPureSparkV1Task.scala:973-1017
  Inferencer reads them in batches of `inference_batch_size`, collates, calls `infer_batch`, emits a row per root
      python/gigl/src/inference/v1/gnn_inferencer.py:234-340, v1/lib/utils.py:78-228
  Trainer reads the split generator's files in batches of `main_sample_batch_size`
      python/gigl/src/common/modeling_task_specs/node_classification_modeling_task_spec.py:134-173
  in-memory precedent inside the reference (graph partitioned by `node_id % world`, batches sampled per rank):
      python/gigl/distributed/distributed_neighborloader.py:26-192,
      python/gigl/distributed/dist_link_prediction_data_partitioner.py:666-714

What stays identical to the TFRecord route (gigl_amd/inferencer.py `_run_tfrecord`): the ROOT ORDER and the BATCH
COMPOSITION.  A root's embedding depends on its batch (the reference runs every layer over the batch's union graph:
a node that two samples share aggregates the in-edges of both), so the in-HBM route walks the roots exactly as the
TFRecord route would read them — part files of config.RECORDS_PER_PART_FILE records in node-id order, the file list permuted
once by RandomState(42) (tf_records_iterable_dataset.py:66-68), records batched in order across file boundaries —
without any of those files existing.  tests/test_gpu_hbm_route.py: both routes give the same rows (1e-5).

WORLD_SIZE > 1: rank r holds the CSC rows and feature rows of the nodes with id % world == r
(gigl_graph_build_shard_from_coo) and runs the sharded plan (gigl_dist_plan_*: per-hop all-to-all, feature pull);
batch c of the global order goes to rank c % world, every rank issues the same number of steps.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .config import GbmlConfigPbWrapper


def part_file_names(prefix: str, n_records: int, records_per_file: Optional[int] = None) -> List[str]:
    """the part files subgraph_sampler._PartWriter creates under `prefix` for n_records records (an empty dataset
    still leaves one empty part file)"""
    from . import config
    records_per_file = int(records_per_file or config.RECORDS_PER_PART_FILE)
    is_dir = prefix.endswith("/") or prefix.endswith(os.sep)
    k = max(1, -(-n_records // records_per_file))
    return [(os.path.join(prefix, f"part-{i:05d}.tfrecord") if is_dir else f"{prefix}{i:05d}.tfrecord")
            for i in range(k)]


def planned_root_order(node_ids: np.ndarray, prefix: str, records_per_file: Optional[int] = None,
                       seed: int = 42) -> np.ndarray:
    """the order in which iterate_tfrecord_batches(tfrecord_files(prefix), ...) would read the per-node records
    that SubgraphSampler writes for `node_ids` (ascending) under `prefix` — computed, no file is touched"""
    from . import config
    from .batches import permuted_files
    records_per_file = int(records_per_file or config.RECORDS_PER_PART_FILE)
    node_ids = np.asarray(node_ids)
    names = sorted(part_file_names(prefix, int(node_ids.size), records_per_file))
    first = {nm: i * records_per_file for i, nm in enumerate(names)}
    order = [node_ids[first[nm]: first[nm] + records_per_file] for nm in permuted_files(names, seed)]
    return np.concatenate(order) if order else node_ids[:0]


@dataclass
class HbmRootBatch:
    """what the in-HBM route hands to `infer_batch` / the training loop in place of a collated TFRecord batch:
    `groups` consecutive batches of `group_roots` roots each (independent batches: dedup, union graph and message
    passing never cross a group boundary), already padded to whole batches.  Padding repeats the batch's first root —
    a repeated root adds no node and no edge to the batch's union graph — and padded rows are dropped from results."""
    resident: "ResidentGraph"
    roots: torch.Tensor             # int32 [groups * group_roots] on the device (uint32 ids)
    group_roots: int
    lane: int = field(default=0, kw_only=True)  # which of the resident graph's ctx / plan sets encodes it (infer_resident)
    valid: Optional[torch.Tensor]   # int64 [n_valid] on the device: positions of the real roots in `roots`; None = all
    root_ids: np.ndarray            # int64 [n_valid] host: the real roots, in order
    root_node_labels: Optional[torch.Tensor] = None  # int64 [n_valid]
    node_type: str = "node"

    @property
    def root_nodes(self):
        from .batches import Node
        return [Node(type=self.node_type, id=int(v)) for v in self.root_ids.tolist()]

    @property
    def groups(self) -> int:
        return int(self.roots.numel()) // self.group_roots

    @property
    def root_ids_dev(self) -> torch.Tensor:
        """the real roots as int64 ids on the device (what the device-side Avro encoder takes: no upload per call)"""
        ids = self.roots.to(torch.int64) & 0xFFFFFFFF
        return ids if self.valid is None else ids.index_select(0, self.valid)


@dataclass
class HbmTrainBatch:
    """a training batch sampled in HBM: the fields the node-classification loop reads from
    SupervisedNodeClassificationBatch (supervised_node_classification_data_loader.py:32-41); `graph` is the
    device-resident HipBatch (sampled trees + union graph), `graph.to(device)` is the identity"""
    graph: "object"
    root_node_indices: torch.Tensor
    root_node_labels: Optional[torch.Tensor]
    root_ids: np.ndarray

    @property
    def root_nodes(self):
        from .batches import Node
        return [Node(type="node", id=int(v)) for v in self.root_ids.tolist()]


@dataclass
class HbmNablpBatch:
    """a link-prediction training batch sampled in HBM: what `infer_task_inputs` reads from a collated
    NodeAnchorBasedLinkPredictionBatch (node_anchor_based_link_prediction_data_loader.py:33-60) without the samples
    ever becoming TFRecords.  The batch graph is the union of the k-hop trees of every anchor AND of every sampled
    positive — the node / edge set the TFRecord route's collate builds from the records' merged neighbourhoods
    (NodeAnchorBasedLinkPredictionTask.scala:146-312) — rooted at [anchor, pos_1 .. pos_P] per anchor; a positive
    the anchor does not have repeats the anchor (a repeated root adds nothing to the union graph)."""
    graph: "object"                # HipBatch over the n_anchors * trees_per_anchor roots
    n_anchors: int
    trees_per_anchor: int          # 1 + numPositiveSamples
    anchor_ids: np.ndarray         # int64 [n_anchors] host
    n_pos: np.ndarray              # int64 [n_anchors] host: sampled positives per anchor (>= 1)
    pos_rows: torch.Tensor         # int64 [sum(n_pos)] device: positions of the real positives in the root list, anchor-major
    root_ids: torch.Tensor         # int64 [n_anchors * trees_per_anchor] device: global ids of the root list
    root_index: Optional[torch.Tensor] = None  # int64 [n_anchors * trees_per_anchor] device: the roots' rows of model(graph)

    @property
    def root_nodes(self):
        from .batches import Node
        return [Node(type="node", id=int(v)) for v in self.anchor_ids.tolist()]


class ResidentGraph:
    """the job's graph + node features in HBM, with the plans that run batches over it.  One per process (rank)."""

    def __init__(self, cfg: GbmlConfigPbWrapper, device: torch.device, rank: int = 0, world: int = 1, group=None,
                 sampling_seed: Optional[int] = None, need_out_graph: bool = False, sharded: Optional[bool] = None):
        """sharded (default: world > 1): rank r holds the rows of the nodes with id % world == r and batches run
        through the sharded plan; sharded=False keeps a replica of the whole graph on every rank (graphs that fit one
        GPU: no data-path collective, the ranks only split the batches — what DDP training uses)"""
        from ._lib import MODE_REPLACE, MODE_SPARK_HASH
        from .engine import HipEngine
        from .subgraph_sampler import load_preprocessed_graph
        if cfg.is_heterogeneous:
            raise NotImplementedError("the in-HBM route covers homogeneous graphs; typed graphs take the TFRecord route")
        self.cfg, self.device, self.rank, self.world, self.group = cfg, torch.device(device), int(rank), int(world), group
        self.fanouts = [int(f) for f in cfg.fanouts]
        # same seed / mode rule as SubgraphSampler.run
        self.mode = (MODE_REPLACE if str(cfg.experimental_flags.get("sample_with_replacement", "false")).lower() == "true"
                     else MODE_SPARK_HASH)
        seed = 42
        if cfg.permutation_strategy != "deterministic" and sampling_seed is None:
            seed = 1 + int.from_bytes(os.urandom(3), "little") % ((1 << 20) - 1)
            if world > 1:  # one seed for the job
                import torch.distributed as dist
                box = [seed]
                dist.broadcast_object_list(box, src=0, group=group)
                seed = int(box[0])
        self.seed = int(sampling_seed if sampling_seed is not None else seed)
        n, src, dst, x, labels, node_ids = load_preprocessed_graph(cfg)
        self.n = int(n)
        self.node_ids = np.asarray(node_ids, dtype=np.int64)
        self.labels = labels
        directed = bool(cfg.is_graph_directed)
        efeat = getattr(cfg, "edge_features", None)
        multi = bool(directed and efeat is None)
        # in-degree > 0 (createSupervisedNodeClassificationSubgraph emits labeled samples only for such roots)
        deg = np.bincount(dst.astype(np.int64), minlength=n)
        if not directed:
            deg = deg + np.bincount(src.astype(np.int64), minlength=n)
        self.has_in_edge = deg[: n] > 0
        self.engine = eng = HipEngine(self.device.index or 0)
        self.comm = None
        self._plans: Dict[tuple, object] = {}
        self._unsettled: dict = {}   # overflow-flag slots of pipelined calls nobody has settled yet (call_overflowed), by id
        self.overflow_redone = 0     # plan calls redone through the staged launches (their batch outgrew the workspace)
        self.sharded = bool(self.world > 1 if sharded is None else (sharded and self.world > 1))
        if self.sharded and self.mode == MODE_REPLACE:
            # (the sharded plan samples duplicate-free trees; refused HERE, before anything is built, so that a caller
            # on the auto route can still fall back to the sampler's files)
            raise NotImplementedError("sample_with_replacement on a hash-partitioned graph: use the TFRecord route")
        if not self.sharded:
            eng.build_from_coo(n, src, dst, is_directed=directed, keep_multi_edges=multi)
            if need_out_graph:
                eng.build_from_coo(n, dst, src, is_directed=directed, out_graph=True, keep_multi_edges=multi)
            eng.load_features(x)
            if efeat is not None:
                eng.load_edge_features(src, dst, efeat, directed)
        else:
            if efeat is not None:
                raise NotImplementedError("edge features on a hash-partitioned graph: use the TFRecord route")
            eng.build_shard_from_coo(n, self.rank, self.world, src, dst, is_directed=directed, keep_multi_edges=multi)
            if need_out_graph:
                # the SUPERVISION edges (a link-prediction job draws an anchor's positives from its out-edges: counter 3) stay
                # a replica on every rank — 4 bytes per edge, 7 GB at MAG240M — while the message-passing graph (in-edge
                # rows) and the feature rows, the 375 GB, are the partitioned part: any rank can draw the positives of any
                # anchor without an exchange
                eng.build_from_coo(n, dst, src, is_directed=directed, out_graph=True, keep_multi_edges=multi)
            eng.load_features(np.ascontiguousarray(x[self.rank:: self.world]))
            # every hash window of the job ends below (hops + 1) * n + seed * hops + max degree (an upper bound of the
            # degree is enough): lets the owners serve every request from the threshold table
            bound = (len(self.fanouts) + 1) * n + self.seed * len(self.fanouts) + int(deg.max() if deg.size else 0)
            self.max_window_end = bound if bound < (1 << 30) else -1
            self.comm = self._make_comm()
        self.feat_dim = int(x.shape[1])
        self.node_type = str(cfg.node_types[0])
        if cfg.task_kind == "node_classification":
            self._order_prefixes = [cfg.unlabeled_tfrecord_uri_prefix]
        else:
            self._order_prefixes = list(cfg.random_negative_tfrecord_uri_prefixes.values())

    @classmethod
    def from_engine(cls, engine, node_ids: np.ndarray, fanouts: Sequence[int], *, node_type: str = "node",
                    order_prefix: str = "unlabeled/samples/", sampling_seed: int = 42, mode: int = 0) -> "ResidentGraph":
        """a resident graph over an engine that already holds the graph and the feature table (built in HBM by the
        caller — bench.py's synthetic workloads — instead of read from the preprocessor's tables); single rank"""
        self = cls.__new__(cls)
        self.cfg, self.device, self.rank, self.world, self.group = None, engine.device, 0, 1, None
        self.fanouts, self.seed, self.mode = [int(f) for f in fanouts], int(sampling_seed), int(mode)
        self.n, self.node_ids, self.labels = int(engine.n_nodes), np.asarray(node_ids, dtype=np.int64), {}
        self.has_in_edge = None
        self.engine, self.comm, self._plans, self.sharded = engine, None, {}, False
        self._unsettled, self.overflow_redone = {}, 0
        self.feat_dim, self.node_type, self._order_prefixes = int(engine.feat_dim), node_type, [order_prefix]
        self._borrowed_engine = True
        return self

    # ---- multi-GPU transport
    def _make_comm(self):
        from .dist import Comm
        return Comm.from_torch(self.engine, self.group)

    # ---- root order
    def inference_root_order(self) -> np.ndarray:
        """every node, in the order the TFRecord route reads the sampler's per-node records (the unlabeled
        RootedNodeNeighborhood samples of a node-classification job; the random-negative stream of a link-prediction
        job: v1/lib/utils.py:78-228)"""
        ids = [planned_root_order(self.node_ids, p) for p in self._order_prefixes if p]
        return np.concatenate(ids) if ids else self.node_ids

    def labeled_root_order(self, label_key: Optional[str] = None) -> Tuple[np.ndarray, np.ndarray]:
        """(roots, labels) of the labeled training samples in the order SubgraphSampler writes them: ascending ids that
        have a label and at least one in-edge, capped by numMaxTrainingSamplesToOutput"""
        pm = self.cfg.preprocessed_metadata.nodes[0]
        key = label_key or (pm.label_keys[0] if pm.label_keys else None)
        lab = self.labels.get(key, {}) if key else {}
        ids = np.array([i for i in self.node_ids.tolist() if i in lab and self.has_in_edge[i]], dtype=np.int64)
        limit = self.cfg.num_max_training_samples_to_output
        if limit > 0:
            ids = ids[:limit]
        return ids, np.array([lab[i] for i in ids.tolist()], dtype=np.int64)

    # ---- batches
    def root_batches(self, ids: np.ndarray, batch_size: int, groups: int = 1,
                     labels: Optional[np.ndarray] = None, shard: bool = True) -> Iterator[HbmRootBatch]:
        """consecutive batches of `batch_size` roots, `groups` per HbmRootBatch.  WORLD_SIZE > 1 (shard=True): batch c
        goes to rank c % world and every rank yields the same number of HbmRootBatches (the sharded plan's exchanges
        are collective); a rank without real batches left yields all-padding ones"""
        ids = np.asarray(ids, dtype=np.int64)
        b, g = int(batch_size), max(1, int(groups))
        n_batches = -(-ids.size // b) if ids.size else 0
        world, rank = (self.world, self.rank) if shard else (1, 0)
        per_rank = -(-n_batches // world) if n_batches else 0
        calls = -(-per_rank // g) if per_rank else 0
        if calls == 0:
            return
        node_type = self.node_type
        # every root of this rank's calls in ONE padded array and ONE upload (a pageable upload per call would wait
        # for the stream and serialise the pipeline): slot (c, k) holds batch rank + (c*g + k)*world of the global order
        slot_batch = rank + np.arange(calls * g, dtype=np.int64) * world          # global batch index per slot
        real_slot = slot_batch < n_batches
        src = np.minimum(slot_batch, max(n_batches - 1, 0))[:, None] * b + np.arange(b)[None, :]  # [slots, b] positions
        in_range = (src < ids.size) & real_slot[:, None]
        first = np.where(real_slot, np.minimum(slot_batch, max(n_batches - 1, 0)) * b, 0)  # a batch's first root pads it
        roots = np.where(in_range, ids[np.minimum(src, ids.size - 1)], ids[first][:, None])   # (all-padding: ids[0])
        roots_dev = torch.from_numpy(roots.astype(np.uint32).view(np.int32).reshape(calls, g * b)).pin_memory() \
            .to(self.device, non_blocking=True)
        valid_mask = in_range.reshape(calls, g * b)
        labels = None if labels is None else np.asarray(labels)
        # positions of the real roots of the calls that carry padding, uploaded before the first call runs
        partial = [c for c in range(calls) if not valid_mask[c].all()]
        pos_dev = {c: torch.from_numpy(np.flatnonzero(valid_mask[c])).to(self.device) for c in partial}
        for c in range(calls):
            m = valid_mask[c]
            pos = np.flatnonzero(m)
            real_pos = src.reshape(calls, g * b)[c][pos]
            full = pos.size == g * b
            yield HbmRootBatch(
                resident=self, roots=roots_dev[c], group_roots=b,
                valid=None if full else pos_dev[c],
                root_ids=ids[real_pos],
                root_node_labels=(torch.from_numpy(labels[real_pos]) if labels is not None else None),
                node_type=node_type)

    # ---- forward over a HbmRootBatch
    def lane_engine(self, lane: int):
        """the engine (ctx + scratch + stream binding) of inference lane `lane`: lane 0 is the resident graph's own, the
        others borrow its HBM-resident graph and table (HipEngine.share_resident) — several calls of the one-call plan
        can then be in flight on streams of their own, one's sampler / union under another's layers, like bench.py's"""
        if lane == 0 or self.sharded:
            return self.engine
        lanes = self.__dict__.setdefault("_lane_engines", {})
        if lane not in lanes:
            from .engine import HipEngine
            e = HipEngine(self.device.index or 0)
            e.share_resident(self.engine)
            lanes[lane] = e
        return lanes[lane]

    def _plan_for(self, model, b: int, groups: int, lane: int = 0):
        """the one-call plan of `model` for `groups` batches of b roots (None when the model / world has none) on inference
        lane `lane`; weights are refreshed from the model at every use (they may have been trained since)"""
        key = (id(model), int(b), int(groups)) if lane == 0 else (id(model), int(b), int(groups), int(lane))
        plan = self._plans.get(key)
        if plan is None and key not in self._plans:
            plan = self._build_plan(model, b, groups, self.lane_engine(lane))
            self._plans[key] = plan
        if plan is not None:
            # weights are snapshotted by the plan: refresh them when a parameter has changed since (in-place updates by
            # the optimiser bump `_version`; re-assigned tensors change the pointer)
            stamp = tuple((p.data_ptr(), p._version) for p in model.parameters())
            if getattr(plan, "_weights_stamp", None) != stamp:
                self._refresh(plan, model)
                plan._weights_stamp = stamp
        return plan

    def _build_plan(self, model, b: int, groups: int, eng=None):
        eng = eng if eng is not None else self.engine
        if self.sharded:
            from .dist import DistSagePlan
            from .models import GraphSAGE
            from .models_attn import GAT
            if type(model) is GAT:  # (raises NotImplementedError for options outside the sharded plan)
                return model.make_dist_plan(self.comm, groups * b, self.fanouts, group_roots=b,
                                            max_window_end=self.max_window_end)
            if type(model) is not GraphSAGE or not model._plain or model.aggr not in ("mean", "sum", "max") or \
                    model.feats_interaction is not None or model.feature_embedding_layer is not None:
                if encoder_trains_over_graph_data(model):
                    return None  # every other encoder of the zoo: staged batches (graph_data) + the encoder's own forward
                raise NotImplementedError("WORLD_SIZE > 1: the sharded route runs the package's encoders (got "
                                          f"{type(model).__name__})")
            w, bs = model.fused_params()
            # rows wider than the first layer's output: project this rank's shard once, pull W_l x rows (the table is
            # refilled in place by _refresh when the weights change)
            proj = None
            if len(self.fanouts) == 2 and model.aggr != "max" and model.projected_input_pays(self.engine) and \
                    os.environ.get("GIGL_AMD_PROJECT_INPUT", "1") != "0":
                cache = self.__dict__.setdefault("_dist_proj", {})  # one projected shard per model
                proj = cache.get(id(model))
                if proj is None:
                    proj = cache[id(model)] = self.engine.project_features(w[0])
            return DistSagePlan(self.comm, w, bs, groups * b, self.fanouts, act_last=model.activation_after_last_conv,
                                group_roots=b, max_window_end=self.max_window_end, projected=proj, aggr=model.aggr)
        make = getattr(model, "make_plan", None)
        if make is None:
            return None
        from ._lib import MODE_REPLACE, GiglError
        if self.mode == MODE_REPLACE:
            return None  # the one-call plan needs duplicate-free trees: staged sample -> union -> forward below
        try:
            return make(eng, b, self.fanouts, groups=groups)
        except NotImplementedError:
            return None  # options outside the one-call plan: staged forward below
        except GiglError as e:
            if e.code != -4:  # GIGL_E_UNSUPPORTED: a shape the plan's kernels are not built for -> staged forward
                raise
            return None

    def _refresh(self, plan, model) -> None:
        if hasattr(model, "fused_params"):
            plan.set_weights(*model.fused_params())
            if self.sharded and getattr(plan, "projected", None) is not None:  # same table, new weights
                self.engine.project_features(model.conv_layers[0].fused_weight(), out=plan.projected)
            # an inference pass over rows wider than the first layer's output: project the table once per model state
            # (X W_l^T, X W_r^T) and run the first layer over projected rows — one table per resident graph, shared by
            # the plans of this model
            if not self.sharded and hasattr(plan, "set_projected_input") and \
                    getattr(model, "projected_input_pays", lambda e: False)(self.engine) and \
                    os.environ.get("GIGL_AMD_PROJECT_INPUT", "1") != "0":
                stamp = tuple((p.data_ptr(), p._version) for p in model.conv_layers[0].parameters())
                cached = getattr(self, "_proj_tables", None)
                if cached is None or cached[0] != (id(model), stamp):
                    w0 = model.conv_layers[0].fused_weight()
                    old = cached[1] if cached is not None and cached[1].shape[1] == 2 * w0.shape[0] else None
                    cached = self._proj_tables = ((id(model), stamp), self.engine.project_features(w0, out=old))
                    self.engine.synchronize()  # (once per model state: the other lanes' plans read the table too)
                plan.set_projected_input(cached[1])
        elif hasattr(model, "plan_params"):
            plan.set_weights(*model.plan_params())

    def encode(self, model, batch: HbmRootBatch) -> torch.Tensor:
        """root embeddings [n_valid, out] of the batch's real roots, in order (inference: no autograd)"""
        lane = int(getattr(batch, "lane", 0))
        eng = self.lane_engine(lane)
        if lane:
            eng.bind_stream()  # (the caller runs this lane under a stream of its own: a no-op once bound)
        eng.bind_stream(torch.cuda.current_stream(self.device))
        b, g = batch.group_roots, batch.groups
        with torch.no_grad():
            plan = None if getattr(batch, "force_staged", False) else self._plan_for(model, b, g, lane)
            if plan is not None:
                # A call whose batch did not fit the one-call plan's workspace (roots that are each other's sampled
                # neighbours turn whole leaf rows into inner rows: common on graphs of a few thousand nodes, never seen on
                # large ones) hands out NaN rows and sets a device flag.  Such a call is REDONE batch by batch through the
                # staged launches below, whose buffers are sized from the batch's own counts — the reference's collate has
                # no workspace to overflow (rooted_node_neighborhood_data_loader.py:78-158).  The flag is read here, unless
                # the caller pipelines its calls (batch.defer_overflow_check: Inferencer.infer_resident keeps a few calls
                # in flight and settles each one — call_overflowed — before its rows are handed on).
                if self.sharded:
                    out = plan.run(batch.roots, sampling_seed=self.seed)
                    # every rank takes part in a redo (the staged route's exchanges are collectives): the flags are reduced
                    redo = self._any_rank(plan.overflowed())
                    if not redo and getattr(model, "should_l2_normalize_embedding_layer_output", False) and \
                            type(plan).__name__ == "DistSagePlan":
                        out = torch.nn.functional.normalize(out, p=2, dim=1)  # (the encoder's last step: row-wise)
                else:
                    out = plan.run(batch.roots, sampling_seed=self.seed, mode=self.mode)
                    slot = self._overflow_slot()
                    plan.overflow_add(slot["dev"])
                    if getattr(batch, "defer_overflow_check", False):
                        slot["host"].copy_(slot["dev"], non_blocking=True)
                        slot["event"].record(torch.cuda.current_stream(self.device))
                        batch._overflow_slot = slot
                        self._unsettled[id(slot)] = slot
                        redo = False
                    else:
                        redo = bool(int(slot["dev"].item()))
                        slot["busy"] = False
                if not redo:
                    return out if batch.valid is None else out.index_select(0, batch.valid)
                self.overflow_redone += 1
            outs = []
            as_graph_data = self.sharded or not encoder_takes_hip_batches(model)
            if as_graph_data and getattr(model, "engine", None) is None:
                model.engine = self.engine  # (a GraphData batch carries no engine of its own)
            for k in range(g):  # staged: sample -> union -> model(HipBatch), one batch at a time
                if as_graph_data:  # (a sharded graph: the batch assembled from the ranks' shards; encoders without a
                    # forward over HipBatches: the same batch as a PyG-shaped GraphData built on the device)
                    # (wide: hop / row buckets at their worst-case sizes — this loop is where a call that overflowed the
                    # one-call sharded plan is redone, on every rank alike)
                    gd, ri = self.graph_data(batch.roots[k * b: (k + 1) * b].contiguous(), pad_to=b, wide=self.sharded)
                    outs.append(model(gd)[ri])
                    continue
                hb = self.hip_batch(batch.roots[k * b: (k + 1) * b])
                outs.append(model(hb)[hb.root_local.long()])
            out = torch.cat(outs)
            return out if batch.valid is None else out.index_select(0, batch.valid)

    def _any_rank(self, flag: bool) -> bool:
        if self.world <= 1:
            return bool(flag)
        import torch.distributed as dist
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if dist.get_backend(self.group) == "nccl":
            t = t.to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    def _overflow_slot(self) -> dict:
        """a (device int32, pinned host int32, event) triple for one call's overflow flag, from a small ring"""
        ring = self.__dict__.setdefault("_overflow_ring", [])
        for sl in ring:
            if not sl["busy"]:
                break
        else:
            sl = dict(dev=torch.zeros(1, dtype=torch.int32, device=self.device),
                      host=torch.zeros(1, dtype=torch.int32).pin_memory(), event=torch.cuda.Event(), busy=False)
            ring.append(sl)
        sl["busy"] = True
        sl["dev"].zero_()
        return sl

    def call_overflowed(self, batch) -> bool:
        """settle a call issued with batch.defer_overflow_check: waits for that call (long finished when the caller keeps a
        few calls in flight) and says whether its rows are NaN — the caller then encodes the batch again with
        batch.force_staged = True (the staged launches) before it hands the rows on"""
        sl = getattr(batch, "_overflow_slot", None)
        if sl is None:
            return False
        batch._overflow_slot = None
        sl["event"].synchronize()
        over = bool(int(sl["host"][0]))
        sl["busy"] = False
        self._unsettled.pop(id(sl), None)
        return over

    def raise_on_overflow(self) -> None:
        """RuntimeError when a deferred call (batch.defer_overflow_check) overflowed and was never settled through
        call_overflowed: its NaN rows were handed on.  Synchronises; callers check once per pass, before the rows are
        declared written.  (Calls that are not deferred are checked — and redone — inside encode.)"""
        n = 0
        for sl in list(self._unsettled.values()):
            sl["event"].synchronize()
            n += int(sl["host"][0])
            sl["busy"] = False
        self._unsettled.clear()
        if n:
            raise RuntimeError(f"{n} pipelined plan call(s) overflowed their batch workspace and were not settled "
                               "(ResidentGraph.call_overflowed): their rows are NaN")

    def hip_batch(self, roots: torch.Tensor, train: bool = False):
        """sampled trees + the batch union graph of `roots` (int32 device ids) as a models.HipBatch"""
        from .models import HipBatch
        if self.sharded:
            raise NotImplementedError("staged batches on a hash-partitioned graph: use the sharded plan (encode)")
        eng = self.engine
        eng.bind_stream(torch.cuda.current_stream(self.device))  # (a no-op unless the caller's stream changed)
        tree = eng.sample_khop(roots, self.fanouts, sampling_seed=self.seed, mode=self.mode)
        u = eng.union_build(tree)
        return HipBatch(eng, tree, u, train=train)

    # encoders without an autograd forward over HipBatches (GAT, GCN, GIN, Transformer, GraphSAGE with batch norm / JK ...)
    # train over the same in-HBM batch as a GraphData built on the device (set by the task specs)
    train_as_graph_data: bool = False
    defer_x: bool = False  # graph_data leaves the dense feature matrix out (GraphData.x_fn builds it on demand)

    def graph_data(self, roots: torch.Tensor, pad_to: Optional[int] = None, wide: bool = False):
        """the batch of `roots` (int32 device ids) as a nn.GraphData on the device — x = the union nodes' feature rows,
        edge_index = the batch union graph's distinct edges (src -> dst, local ids), edge_attr when the job has edge
        features — what the trainer-side collate builds from the samples' records (pyg_graph_builder.py:20-69), for the
        encoders that train over a PyG-shaped batch; one host read (the batch's node / edge counts) -> (graph, root rows)"""
        from .nn import GraphData
        eng = self.engine
        eng.bind_stream(torch.cuda.current_stream(self.device))
        dev = self.device
        n_real = int(roots.numel())
        if self.sharded:
            # the graph is hash-partitioned over the ranks: the per-hop requests, the union graph and the feature pull of
            # the sharded plan (a STAGED plan: every union node numbered, raw rows), then the batch as it stands in this
            # rank's HBM.  Every rank calls this once per step with the same batch size (short batches are padded with
            # their first root: a repeated root adds nothing to the union graph).
            b = int(pad_to or n_real)
            if n_real < b:
                roots = torch.cat([roots, roots[:1].expand(b - n_real)]).contiguous()
            plan = self._staged_plan(b, wide)
            plan.sample_and_pull(roots, sampling_seed=self.seed)
            t = plan.batch_tensors()
            m = t["meta"].cpu().tolist()
            if self._any_rank(bool(m[8])) and not wide:
                # a hop / feature-row bucket of the default sizes overflowed on some rank: every rank redoes the batch with
                # the buckets at their worst-case sizes (hop buckets = the whole frontier, row buckets = every union node)
                return self.graph_data(roots[:n_real], pad_to=pad_to, wide=True)
            if m[8]:
                raise RuntimeError("sharded batch failed with worst-case buckets (meta[GIGL_META_OVERFLOW]): not a capacity")
            n, e = int(m[0]), int(m[1])
            rowptr, rowend, col, root_local, x = t["rowptr"], t["rowend"], t["col"], t["root_local"], t["x"][:n]
            node_ids = levels = None  # (the rows were pulled from their owners: no resident table to read in place)
        else:
            tree = eng.sample_khop(roots, self.fanouts, sampling_seed=self.seed, mode=self.mode)
            u = eng.union_build(tree)
            c = u.counts()  # (raises when the batch did not fit its workspace)
            n, e = int(c["n_nodes"]), int(c["n_edges"])
            rowptr, rowend, col, root_local = u.rowptr, u.rowend, u.col, u.root_local
            node_ids, levels = u.nodes[:n], [int(v) for v in c["levels"]]
            # (defer_x: the consumer reads the stored rows in place — models_attn.GAT's input-side training forward — and
            # asks for the dense matrix only if it falls back: GraphData.features())
            x_fn = (lambda nodes=u.nodes, cnt=u.meta[:1], n=n: eng.gather_rows(nodes, cnt, n))
            x = None if self.defer_x else x_fn()
        rp, re = rowptr[:n].to(torch.int64), rowend[:n].to(torch.int64)
        lens = re - rp
        start = torch.cumsum(lens, 0) - lens
        dst = torch.repeat_interleave(torch.arange(n, device=dev), lens, output_size=e)
        idx = torch.repeat_interleave(rp - start, lens, output_size=e) + torch.arange(e, device=dev)
        src = col.index_select(0, idx).to(torch.int64)
        ea = None
        if not self.sharded and getattr(eng, "_efeat", None) is not None:
            ea = eng.union_edge_attr(u).index_select(0, idx)
        g = GraphData(x=x, edge_index=torch.stack([src, dst]), edge_attr=ea)
        if x is None:
            g.x_fn = x_fn
        # the CSR by destination the kernels read IS the union graph's (rows ascending, a row's sources ascending: the
        # order GraphData._build_csr sorts into) — packed here, not rebuilt by a sort of the edge list
        rp = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        rp[1:] = (start + lens).to(torch.int32)
        g.rowptr = rp
        g.col = src.to(torch.int32).contiguous() if e else torch.zeros(1, dtype=torch.int32, device=dev)
        g.n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
        g.edge_attr_csr = ea
        if node_ids is not None and getattr(eng, "_feat", None) is not None:
            g.node_ids, g.table, g.levels = node_ids, eng, levels
        return g, root_local[:n_real].to(torch.int64)

    def _staged_plan(self, b: int, wide: bool = False):
        """the sharded plan that serves training batches of b roots (dist.DistSagePlan(staged=True): its forward is never
        run, the weights are placeholders).  wide: every bucket at its worst-case size — a batch cannot overflow it (the
        plan a batch is redone through when it overflowed the default buckets)"""
        plans = self.__dict__.setdefault("_staged_plans", {})
        if (b, wide) not in plans:
            from .dist import DistSagePlan
            L = len(self.fanouts)
            w = [torch.zeros((4, 2 * (self.feat_dim if l == 0 else 4)), device=self.device) for l in range(L)]
            kw = dict(hop_slack=float(max(self.world, 1)), pull_cap=1 << 40) if wide else {}
            plans[(b, wide)] = DistSagePlan(self.comm, w, [None] * L, b, self.fanouts, max_window_end=self.max_window_end,
                                            staged=True, **kw)
        return plans[(b, wide)]

    def train_graph(self, roots: torch.Tensor, pad_to: Optional[int] = None):
        """-> (what the model's forward takes, the roots' rows of its output) for a training / validation batch;
        pad_to: the job's batch size (a hash-partitioned graph needs every rank's batch at one size)"""
        if self.train_as_graph_data or self.sharded:
            return self.graph_data(roots, pad_to)
        hb = self.hip_batch(roots, train=True)
        return hb, hb.root_local.long()

    def train_batches(self, ids: np.ndarray, labels: np.ndarray, batch_size: int) -> Iterator[HbmTrainBatch]:
        """training batches sampled in HBM: consecutive `batch_size` roots; rank r takes batches r, r + world, ... and
        every rank takes the same number (a short rank wraps around to the first batches: the gradient all-reduce of
        DistributedDataParallel needs equal step counts, cf. the file tiling of data_loaders/utils.py:38-47)"""
        ids = np.asarray(ids, dtype=np.int64)
        n_batches = -(-ids.size // batch_size) if ids.size else 0
        per_rank = -(-n_batches // self.world) if n_batches else 0
        for k in range(per_rank):
            c = (self.rank + k * self.world) % n_batches
            chunk = ids[c * batch_size: (c + 1) * batch_size]
            r32 = torch.from_numpy(chunk.astype(np.uint32).view(np.int32)).to(self.device)
            hb, ri = self.train_graph(r32, pad_to=batch_size)
            yield HbmTrainBatch(graph=hb, root_node_indices=ri,
                                root_node_labels=torch.from_numpy(labels[c * batch_size: c * batch_size + chunk.size]),
                                root_ids=chunk)

    # ---- link-prediction training batches (the homogeneous NABLP trainer's in-HBM route)
    def nablp_anchor_order(self, num_positives: int) -> Tuple[np.ndarray, np.ndarray]:
        """(anchors, positives per anchor) of the main samples in the order the TFRecord route reads them: the sampler
        writes one NodeAnchorBasedLinkPredictionSample per node with at least one sampled positive and a neighbourhood of
        its own, in ascending id order, capped by numMaxTrainingSamplesToOutput (subgraph_sampler._run_nablp;
        NodeAnchorBasedLinkPredictionTask.scala:186-194), into part files that are read in permuted order"""
        ids = self.node_ids
        cnt = np.zeros(self.n, dtype=np.int64)
        step = 1 << 20
        for lo in range(0, ids.size, step):  # (one pass over the nodes at setup; the positives themselves are re-drawn per batch)
            chunk = ids[lo:lo + step]
            r32 = torch.from_numpy(chunk.astype(np.uint32).view(np.int32)).to(self.device)
            _, c = self.engine.sample_positives(r32, num_positives, sampling_seed=self.seed)
            cnt[chunk] = c.cpu().numpy()
        emit = ids[(cnt[ids] > 0) & self.has_in_edge[ids]]
        limit = self.cfg.num_max_training_samples_to_output
        if limit > 0:
            emit = emit[:limit]
        order = planned_root_order(emit, self.cfg.nablp_tfrecord_uri_prefix)
        return order, cnt[order]

    def nablp_batches(self, ids: np.ndarray, n_pos: np.ndarray, batch_size: int, num_positives: int,
                      loop: bool = False) -> Iterator[HbmNablpBatch]:
        """consecutive batches of `batch_size` anchors (`ids`, with `n_pos` positives each) sampled in HBM"""
        from itertools import cycle
        P, T = int(num_positives), 1 + int(num_positives)
        spans = [(lo, min(lo + batch_size, ids.size)) for lo in range(0, ids.size, batch_size)]
        if self.world > 1 and spans:
            # rank r takes batches r, r + world, ...; every rank the same number (a short rank wraps around: the gradient
            # all-reduce of DistributedDataParallel — and a hash-partitioned graph's exchanges — need equal step counts)
            per_rank = -(-len(spans) // self.world)
            spans = [spans[(self.rank + k * self.world) % len(spans)] for k in range(per_rank)]
        ar = torch.arange(P, device=self.device).view(1, P)
        for lo, hi in (cycle(spans) if (loop and spans) else spans):
            chunk, k = ids[lo:hi], n_pos[lo:hi]
            anchors = torch.from_numpy(chunk.astype(np.uint32).view(np.int32)).to(self.device)
            pos, cnt = self.engine.sample_positives(anchors, P, sampling_seed=self.seed)
            a2 = anchors.view(-1, 1)
            grouped = torch.where(ar < cnt.view(-1, 1), pos.view(-1, P), a2.expand(-1, P))
            roots = torch.cat([a2, grouped], dim=1).reshape(-1).contiguous()
            hb, ri = self.train_graph(roots, pad_to=batch_size * T)
            k64 = np.asarray(k, dtype=np.int64)
            # rows of the positives in the anchor-major root list: i * T + 1 + j for j < k[i]
            rows = (np.repeat(np.arange(k64.size, dtype=np.int64) * T + 1 - (np.cumsum(k64) - k64), k64)
                    + np.arange(int(k64.sum()), dtype=np.int64))
            yield HbmNablpBatch(graph=hb, n_anchors=int(chunk.size), trees_per_anchor=T, anchor_ids=chunk,
                                n_pos=np.asarray(k, dtype=np.int64),
                                pos_rows=torch.from_numpy(rows.astype(np.int64)).to(self.device),
                                root_ids=roots.to(torch.int64) & 0xFFFFFFFF, root_index=ri)

    def nablp_root_batches(self, ids: np.ndarray, n_pos: np.ndarray, batch_size: int, num_positives: int):
        """the ROOT side of nablp_batches only — (main roots anchor-major, positives per anchor, anchor ids) per batch, no
        batch graph built: what the library's link-prediction training plan takes (engine.NablpTrainPlan samples and
        collates inside its step); same anchors, positives and order as nablp_batches"""
        P = int(num_positives)
        spans = [(lo, min(lo + batch_size, ids.size)) for lo in range(0, ids.size, batch_size)]
        ar = torch.arange(P, device=self.device).view(1, P)
        for lo, hi in spans:
            chunk = ids[lo:hi]
            anchors = torch.from_numpy(chunk.astype(np.uint32).view(np.int32)).to(self.device)
            pos, cnt = self.engine.sample_positives(anchors, P, sampling_seed=self.seed)
            a2 = anchors.view(-1, 1)
            grouped = torch.where(ar < cnt.view(-1, 1), pos.view(-1, P), a2.expand(-1, P))
            yield torch.cat([a2, grouped], dim=1).reshape(-1).contiguous(), cnt.to(torch.int32).contiguous(), chunk

    def random_negative_root_batches(self, batch_size: int):
        """the roots of random_negative_batches, batch by batch, forever (no batch graph built)"""
        order = self.inference_root_order()
        while order.size:
            for lo in range(0, order.size, batch_size):
                chunk = order[lo:lo + batch_size]
                yield torch.from_numpy(chunk.astype(np.uint32).view(np.int32)).to(self.device)

    def random_negative_batches(self, batch_size: int) -> Iterator[HbmTrainBatch]:
        """the random-negative stream of a link-prediction job sampled in HBM: every node's RootedNodeNeighborhood in
        the order the TFRecord route reads the sampler's files, `batch_size` roots per batch (the last batch of a pass
        is short), looping forever like the reference's LoopyIterableDataset (tf_records_iterable_dataset.py:85-109)"""
        order = self.inference_root_order()
        los = list(range(0, order.size, batch_size))
        if self.world > 1 and los:  # (rank r: chunks r, r + world, ... of the pass, the same number on every rank)
            per_rank = -(-len(los) // self.world)
            los = [los[(self.rank + k * self.world) % len(los)] for k in range(per_rank)]
        while order.size:
            for lo in los:
                chunk = order[lo:lo + batch_size]
                r32 = torch.from_numpy(chunk.astype(np.uint32).view(np.int32)).to(self.device)
                hb, ri = self.train_graph(r32, pad_to=batch_size)
                yield HbmTrainBatch(graph=hb, root_node_indices=ri, root_node_labels=None, root_ids=chunk)

    def close(self) -> None:
        for p in self._plans.values():
            if p is not None:
                p.close()
        self._plans = {}
        for e in self.__dict__.pop("_lane_engines", {}).values():
            e.close()
        for p in self.__dict__.pop("_staged_plans", {}).values():
            p.close()
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        if self.engine is not None:
            if not getattr(self, "_borrowed_engine", False):
                self.engine.close()
            self.engine = None


class GraphedTrainStep:
    """one training step on the in-HBM route — k-hop sample + batch union graph + forward with autograd + cross-entropy on
    the roots + backward + the optimiser update — captured ONCE as a HIP graph and replayed per batch.  The step is the
    same sequence of launches the eager loop issues (node_classification_modeling_task_spec.py:134-173 on batches sampled
    in HBM); eager, it is bound by the host issuing ~60 small launches per batch from Python, replayed it is bound by the
    kernels.  Every shape is a capacity (row counts stay on the device), so one capture serves every batch; the root ids
    and labels of a batch are copied into the graph's static inputs.  A batch shorter than `b` is padded with its first
    root (a repeated root adds nothing to the union graph) and masked out of the loss.

    The optimiser must keep its step counter on the device (torch.optim.Adam(..., capturable=True))."""

    def __init__(self, resident: "ResidentGraph", model, optimizer, b: int, warmup_roots: torch.Tensor,
                 warmup_labels: torch.Tensor):
        import copy
        import torch.nn.functional as F
        self.resident, self.model, self.opt, self.b = resident, model, optimizer, int(b)
        dev = resident.device
        self.roots = torch.zeros(self.b, dtype=torch.int32, device=dev)
        self.labels = torch.zeros(self.b, dtype=torch.int64, device=dev)
        self.mask = torch.ones(self.b, dtype=torch.float32, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        eng = resident.engine
        torch.cuda.synchronize(dev)
        eng.bind_stream(self.stream)
        # warm-up iterations (allocator pools, library workspaces, cached constants) must not train: the model and the
        # optimiser are put back afterwards
        saved_model = copy.deepcopy(model.state_dict())
        saved_opt = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                     for p, st in optimizer.state.items()}
        self._set(warmup_roots, warmup_labels)

        def body():
            hb = resident.hip_batch(self.roots, train=True)
            out = model(hb)
            per_row = F.cross_entropy(out[hb.root_local.long()], self.labels, reduction="none")
            loss = (per_row * self.mask).sum() / self.mask.sum()
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            optimizer.step()
            return loss

        try:
            with torch.cuda.stream(self.stream):
                for _ in range(3):
                    body()
            self.stream.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.loss = body()
            self.stream.synchronize()
        finally:
            # whether or not the capture succeeded, the warm-up must not have trained: the model and the optimiser's
            # state tensors (part of the captured graph) go back IN PLACE to what they were — moments and step counter
            # zero for a fresh optimiser — so that an eager fallback starts from the same state
            torch.cuda.synchronize(dev)
            model.load_state_dict(saved_model)
            with torch.no_grad():
                for p, st in optimizer.state.items():
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            old = saved_opt.get(p, {}).get(k)
                            v.copy_(old) if old is not None else v.zero_()
            torch.cuda.synchronize(dev)

    def _set(self, roots: torch.Tensor, labels: torch.Tensor) -> None:
        k = int(roots.numel())
        assert 0 < k <= self.b
        with torch.cuda.stream(self.stream):
            self.roots[:k].copy_(roots.to(torch.int32), non_blocking=True)
            self.labels[:k].copy_(labels.to(self.labels.device, non_blocking=True), non_blocking=True)
            if k < self.b:
                self.roots[k:] = self.roots[0]
                self.labels[k:] = self.labels[0]
            self.mask[:k] = 1.0
            self.mask[k:] = 0.0

    def step(self, roots: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """one optimiser step on the batch (`roots` int32 device ids [k <= b], labels int64 [k]); returns the loss (a
        static device scalar, overwritten by the next step)"""
        self._set(roots, labels)
        with torch.cuda.stream(self.stream):
            self.graph.replay()
        return self.loss


def encoder_takes_hip_batches(model) -> bool:
    """the encoders whose forward runs over a device-resident HipBatch (sampled trees + union graph)"""
    from .models import GraphSAGE
    from .models_attn import GAT, TwoLayerGCN
    return type(model) in (GraphSAGE, GAT, TwoLayerGCN)


def encoder_trains_over_hip_batches(model) -> bool:
    """the encoders with an autograd forward over a HipBatch (models.GraphSAGE._forward_union_autograd)"""
    from .models import GraphSAGE
    return (type(model) is GraphSAGE and not model.batchnorm and model.jk_layer is None
            and model.feats_interaction is None and model.feature_embedding_layer is None)


def encoder_trains_over_graph_data(model) -> bool:
    """the package's homogeneous encoders: every one has an autograd forward over a nn.GraphData batch (what the TFRecord
    route's collate hands them), so the in-HBM route can hand them the same batch built on the device"""
    return type(model).__module__ in ("gigl_amd.models", "gigl_amd.models_attn", "gigl_amd.models_more")


def route_of(cfg: GbmlConfigPbWrapper, args: Dict[str, str], override: Optional[str] = None) -> str:
    """"hbm" | "tfrecord": `override` (the entry point's keyword), else the plugin argument `data_route`, else the
    environment variable GIGL_AMD_ROUTE, else "auto" = in-HBM whenever the job's tables can be read and the graph
    is homogeneous (typed graphs and anything the in-HBM route refuses take the TFRecord route)"""
    want = (override or args.get("data_route") or os.environ.get("GIGL_AMD_ROUTE") or "auto").lower()
    if want not in ("hbm", "tfrecord", "auto"):
        raise ValueError(f"data route {want!r}: expected hbm, tfrecord or auto")
    if want != "auto":
        return want
    if cfg.is_heterogeneous:
        return "tfrecord"
    try:
        pm = cfg.preprocessed_metadata
        from .config import resolve_uri, tfrecord_files
        ok = bool(tfrecord_files(os.path.join(resolve_uri(pm.nodes[0].tfrecord_uri_prefix, cfg.uri_base), ""))) and \
            bool(tfrecord_files(os.path.join(resolve_uri(pm.edges[0].tfrecord_uri_prefix, cfg.uri_base), "")))
    except Exception:  # noqa: BLE001 — no readable tables: the samples must come from TFRecords
        ok = False
    return "hbm" if ok else "tfrecord"
