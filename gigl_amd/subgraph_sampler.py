"""SubgraphSampler — drop-in for the reference component

    python -m gigl.src.subgraph_sampler.subgraph_sampler --job_name --task_config_uri --resource_config_uri
    SubgraphSampler().run(applied_task_identifier, task_config_uri, resource_config_uri, cluster_name=None,
                          debug_cluster_owner_alias=None, custom_worker_image_uri=None,
                          skip_cluster_delete=False, additional_spark35_jar_file_uris=())
        (python/gigl/src/subgraph_sampler/subgraph_sampler.py:73-83)

The reference body creates a Dataproc cluster and submits the Scala/Spark job
(`Main.main(Array(frozenGbmlConfigUri, appName, resourceConfigUri))`, scala/subgraph_sampler/src/main/scala/Main.scala:14-16);
here the same inputs (preprocessed node/edge tf.Example TFRecords named by PreprocessedMetadata) produce the
same outputs (RootedNodeNeighborhood / SupervisedNodeClassificationSample /
NodeAnchorBasedLinkPredictionSample TFRecords under the frozen config's flattenedGraphMetadata URIs) on one GPU.
Cloud-only arguments are accepted and ignored.

Restated reference steps (scala/subgraph_sampler/src/main/scala/libs/task/pureSpark/):
  SupervisedNodeClassificationTask.run :29-124   -> run_node_classification
  NodeAnchorBasedLinkPredictionTask.run :28-144  -> run_node_anchor_link_prediction
  createRootedNodeNeighborhoodSubgraph (SGSPureSparkV1Task.scala:973-1017): EVERY node yields a
      RootedNodeNeighborhood; nodes without in-edges get nodes=[self], edges=[]
  createSupervisedNodeClassificationSubgraph (SupervisedNodeClassificationTask.scala:166-236): labeled
      samples only for roots that have a label and at least one edge
"""
from __future__ import annotations

import argparse
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import wire
from .config import GbmlConfigPbWrapper, tfrecord_files
from .sampler_service import HipKHopSamplerService


def load_preprocessed_graph(cfg: GbmlConfigPbWrapper):
    """loadNodeDataframeIntoSparkSql / loadEdgeDataframeIntoSparkSql (SGSPureSparkV1Task.scala:52-118,120-286):
    node ids are the dense enumerated ids of the Data Preprocessor; features = the featureKeys columns
    concatenated in order (:90-104).  Decoded by the native reader (gigl_amd/ingest.py)."""
    from .ingest import COL_F32, COL_I64, feature_widths, read_columns
    pm = cfg.preprocessed_metadata
    nm, em = pm.nodes[0], pm.edges[0]
    nfiles = tfrecord_files(os.path.join(_res(cfg, nm.tfrecord_uri_prefix), ""))
    cols = [(nm.node_id_key, COL_I64, 1)]
    widths = feature_widths(nfiles[0], nm.feature_keys) if nfiles and nm.feature_keys else []
    cols += [(k, COL_F32, max(w, 1)) for k, w in zip(nm.feature_keys, widths)]
    cols += [(lk, COL_I64, 1) for lk in nm.label_keys if lk not in nm.feature_keys and lk != nm.node_id_key]
    data, cnt = read_columns(nfiles, cols)
    ids = data[nm.node_id_key][:, 0]
    n = int(ids.max()) + 1 if ids.size else 0
    d = int(sum(widths))
    x = np.zeros((n, d), dtype=np.float32)
    if d:
        x[ids] = np.concatenate([data[k][:, :w] for k, w in zip(nm.feature_keys, widths) if w], axis=1)
    labels: Dict[str, Dict[int, int]] = {}
    for lk in nm.label_keys:
        if lk in data and data[lk].dtype == np.int64:
            have = cnt[lk] > 0
            labels[lk] = dict(zip(ids[have].tolist(), data[lk][have, 0].tolist()))
    efiles = tfrecord_files(os.path.join(_res(cfg, em.tfrecord_uri_prefix), ""))
    ecols = [(em.src_node_id_key, COL_I64, 1), (em.dst_node_id_key, COL_I64, 1)]
    # `_edge_features` = the main edge info's featureKeys columns concatenated in order (:172-193); empty without keys
    ewidths = feature_widths(efiles[0], em.feature_keys) if efiles and em.feature_keys else []
    ecols += [(k, COL_F32, max(w, 1)) for k, w in zip(em.feature_keys, ewidths)]
    ed, _ = read_columns(efiles, ecols)
    src = ed[em.src_node_id_key][:, 0].astype(np.uint32)  # ids cast to int32 (:164-168)
    dst = ed[em.dst_node_id_key][:, 0].astype(np.uint32)
    cfg.edge_features = None
    if sum(ewidths):
        cfg.edge_features = np.concatenate([ed[k][:, :w] for k, w in zip(em.feature_keys, ewidths) if w], axis=1)
    return n, src, dst, x, labels, sorted(set(ids.tolist()))


def load_label_edge_table(cfg: GbmlConfigPbWrapper, info):
    """loadEdgeDataframeIntoSparkSql with EdgeUsageType.POS / NEG (SGSPureSparkV1Task.scala:120-216): the user-defined
    label edges and their feature columns; never bidirectionalised (:218-258 applies to MAIN edges only)"""
    from .ingest import COL_F32, COL_I64, feature_widths, read_columns
    em = cfg.preprocessed_metadata.edges[0]
    files = tfrecord_files(os.path.join(_res(cfg, info.tfrecord_uri_prefix), ""))
    cols = [(em.src_node_id_key, COL_I64, 1), (em.dst_node_id_key, COL_I64, 1)]
    widths = feature_widths(files[0], info.feature_keys) if files and info.feature_keys else []
    cols += [(k, COL_F32, max(w, 1)) for k, w in zip(info.feature_keys, widths)]
    ed, _ = read_columns(files, cols)
    src = ed[em.src_node_id_key][:, 0].astype(np.uint32)
    dst = ed[em.dst_node_id_key][:, 0].astype(np.uint32)
    feats = None
    if sum(widths):
        feats = np.concatenate([ed[k][:, :w] for k, w in zip(info.feature_keys, widths) if w], axis=1)
    return src, dst, feats


def _res(cfg: GbmlConfigPbWrapper, uri: str) -> str:
    from .config import resolve_uri
    return resolve_uri(uri, cfg.uri_base)


_STAGE = {}


def _frames_to_host(buf) -> np.ndarray:
    """finished TFRecord frames, device -> host through a reusable pinned staging buffer (the part-file writer consumes
    the bytes before the next batch overwrites them)"""
    import torch
    n = int(buf.numel())
    st = _STAGE.get("buf")
    if st is None or st.numel() < n:
        _STAGE["buf"] = st = torch.empty(max(n, 1 << 22), dtype=torch.uint8, pin_memory=True)
    out = st[:n]
    out.copy_(buf, non_blocking=True)
    torch.cuda.current_stream(buf.device).synchronize()
    return out.numpy()


class _PartWriter:
    """spark-tfrecord writes part files under the prefix directory (TFRecordIO.scala:53-69, overwrite mode);
    takes already framed records (bytes + record offsets) and rolls to a new part every records_per_file"""

    def __init__(self, prefix: str, records_per_file: Optional[int] = None):
        from . import config
        self.prefix, self.per = prefix, int(records_per_file or config.RECORDS_PER_PART_FILE)
        self.is_dir = prefix.endswith("/") or prefix.endswith(os.sep)
        d = prefix if self.is_dir else os.path.dirname(prefix)
        os.makedirs(d or ".", exist_ok=True)
        for old in tfrecord_files(prefix):
            os.remove(old)
        self.files: List[str] = []
        self.fh = None
        self.in_part = 0
        self.n_records = 0

    def _roll(self):
        if self.fh:
            self.fh.close()
        i = len(self.files)
        name = (os.path.join(self.prefix, f"part-{i:05d}.tfrecord") if self.is_dir else f"{self.prefix}{i:05d}.tfrecord")
        self.fh = open(name, "wb")
        self.files.append(name)
        self.in_part = 0

    def add(self, buf: np.ndarray, rec_off: np.ndarray, keep: Optional[np.ndarray] = None) -> None:
        """buf: uint8 frames back to back; rec_off[i]..rec_off[i+1] = record i (empty = skipped); keep: optional mask of
        the records to write"""
        sizes = np.diff(rec_off)
        if keep is not None and not bool(np.all(keep)):  # runs of consecutive kept records, each a contiguous range
            k = np.asarray(keep, dtype=bool) & (sizes > 0)
            edges = np.flatnonzero(np.diff(np.concatenate(([0], k.view(np.int8), [0]))))
            for lo, hi in zip(edges[0::2], edges[1::2]):
                sub = rec_off[lo:hi + 1]
                self.add(buf, sub)
            return
        idx = np.flatnonzero(sizes > 0)
        k = 0
        while k < idx.size:
            if self.fh is None or self.in_part >= self.per:
                self._roll()
            take = min(self.per - self.in_part, idx.size - k)
            lo, hi = int(rec_off[idx[k]]), int(rec_off[idx[k + take - 1] + 1])
            self.fh.write(memoryview(buf[lo:hi]))  # skipped records take 0 bytes: the range is contiguous frames
            self.in_part += take
            self.n_records += take
            k += take

    def add_frame(self, frame: bytes) -> None:
        """one finished TFRecord frame"""
        if self.fh is None or self.in_part >= self.per:
            self._roll()
        self.fh.write(frame)
        self.in_part += 1
        self.n_records += 1

    def close(self) -> List[str]:
        if self.fh is None:
            self._roll()  # an empty dataset still leaves one (empty) part file
        self.fh.close()
        self.fh = None
        return self.files


def _encode_labels(label_keys: Sequence[str], labels: Dict[str, Dict[int, int]], ids: np.ndarray):
    """root_node_labels (field 3 of SupervisedNodeClassificationSample, training_samples_schema.proto:23-27) of
    every root, pre-encoded: the device encoder appends them to the RootedNodeNeighborhood fields"""
    parts, off = [], np.zeros(ids.size + 1, dtype=np.int64)
    for i, nid in enumerate(ids.tolist()):
        b = b"".join(wire._len_delim(3, wire.Label(label_type=lk, label=labels[lk][nid]).SerializeToString())
                     for lk in label_keys if nid in labels.get(lk, {}))
        parts.append(b)
        off[i + 1] = off[i] + len(b)
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy(), off


def sampler_call_size(n_roots: int, fanouts: Sequence[int], feat_dim: int, budget_bytes: int = 4 << 30) -> int:
    """roots per library call of the sampler job (sample -> encode -> frames out).  The device record encoder has a fixed
    ~36 us latency floor per pass: 0.24 of the HBM roofline at 4,096 records per call, 0.35 at 32,768
    (profiles/r05l_encoder_variants.md) — so a call takes as many roots as keep its frames under `budget_bytes`
    (the reference writes 1,000-record part files from whole partitions: SGSPureSparkV1Task.scala:1019-1040; the call size
    here only sets how many records are encoded per launch set), at most 32,768, at least 1,024, never more than there are"""
    nodes, width = 1, 1
    for f in fanouts:
        width *= int(f)
        nodes += width
    per_record = nodes * (int(feat_dim) * 4 + 24) + (nodes - 1) * 12 + 64  # (feature rows dominate; an upper bound)
    call = max(1024, min(32768, (budget_bytes // max(per_record, 1)) // 1024 * 1024))
    return int(max(1, min(call, int(n_roots))))


class SubgraphSampler:
    def run(self, applied_task_identifier: str, task_config_uri: str, resource_config_uri: Optional[str] = None,
            cluster_name: Optional[str] = None, debug_cluster_owner_alias: Optional[str] = None,
            custom_worker_image_uri: Optional[str] = None, skip_cluster_delete: bool = False,
            additional_spark35_jar_file_uris: Sequence[str] = (), *, uri_base: Optional[str] = None,
            device: int = 0, batch_size: Optional[int] = None) -> Dict[str, List[str]]:
        """batch_size: roots per library call (None: sampler_call_size — up to 32,768, where the encoder runs at 0.35 of the
        HBM roofline instead of 0.24 at 4,096)"""
        cfg = GbmlConfigPbWrapper.from_uri(task_config_uri, uri_base=uri_base)
        # experimental_flags.sample_with_replacement: numNeighborsToSample independent uniform draws per parent
        # (sampleWithReplacementUDF, SGSPureSparkV1Task.scala:42-50,355-364; an unseeded java.util.Random there, a
        # counter-based generator keyed by the path here: valid samples, no parity definition)
        from ._lib import MODE_REPLACE, MODE_SPARK_HASH
        self.sampling_mode = (MODE_REPLACE if str(cfg.experimental_flags.get("sample_with_replacement", "false")).lower()
                              == "true" else MODE_SPARK_HASH)
        # permutation_strategy: "deterministic" = the hash permutation with samplingSeed 42 (SamplingStrategy.scala:16-82,
        # reproducible, the parity mode).  Anything else is the reference's F.shuffle (:84-101): a uniformly random
        # permutation with no seed and no parity definition — served by the same kernel under a fresh random seed per
        # job (a different uniform sample every run; seeds < 2^20 keep every hash window inside the range table).
        seed = 42
        if cfg.permutation_strategy != "deterministic":
            seed = 1 + int.from_bytes(os.urandom(3), "little") % ((1 << 20) - 1)
        self.sampling_seed = seed
        if cfg.is_heterogeneous:
            return self._run_graphdb_nablp(cfg, device, seed, int(batch_size or 4096))
        n, src, dst, x, labels, node_ids = load_preprocessed_graph(cfg)
        ids = np.asarray(node_ids, dtype=np.uint32)
        if batch_size is None:
            batch_size = sampler_call_size(ids.size, cfg.fanouts, int(x.shape[1]) if x is not None and x.ndim == 2 else 0)
        self.call_size = int(batch_size)
        with HipKHopSamplerService(n, src, dst, x, cfg.is_graph_directed, device=device, sampling_seed=seed,
                                   keep_multi_edges=getattr(cfg, "edge_features", None) is None) as svc:
            svc.sampling_mode = self.sampling_mode
            if getattr(cfg, "edge_features", None) is not None:  # hydrateEdges: records carry Edge.feature_values
                svc.engine.load_edge_features(src, dst, cfg.edge_features, cfg.is_graph_directed)
            if cfg.task_kind == "node_classification":
                return self._run_node_classification(cfg, svc, ids, labels, batch_size)
            return self._run_nablp(cfg, svc, ids, batch_size)

    def _run_graphdb_nablp(self, cfg: GbmlConfigPbWrapper, device: int, seed: int, batch_size: int):
        """GraphDBNodeAnchorBasedLinkPredictionTask.run (scala_spark35 .../task/graphdb/GraphDBNodeAnchorBasedLinkPredictionTask
        .scala:118-496) on a typed graph: one RootedNodeNeighborhood per node of every anchor / target node type, written
        under nodeTypeToRandomNegativeTfrecordUriPrefix[type]; then for the first supervision edge type one
        NodeAnchorBasedLinkPredictionSample per root of its source type (numMaxTrainingSamplesToOutput caps them), with
        numPositiveSamples sampled positive edges and the positives' neighbourhoods merged in; roots without a positive
        are dropped unless shouldIncludeIsolatedNodesInTraining.  Sampling, assembly, hydration, proto encoding and
        TFRecord framing run on the device."""
        from .graphdb_sampler import EdgeType, HipGraphDBSampler
        if cfg.task_kind != "node_anchor_based_link_prediction":
            raise NotImplementedError("typed graphs: only the node-anchor-based link-prediction sampler task is built")
        sup = cfg.supervision_edge_types
        if not sup:
            raise ValueError("nodeAnchorBasedLinkPredictionTaskMetadata.supervisionEdgeTypes is empty")
        pos_et = EdgeType(*sup[0])  # (stage 1 of the reference: one supervision edge type)
        anchor_target = list(dict.fromkeys([t for e in sup for t in (e[0],)] + [t for e in sup for t in (e[2],)]))
        node_types, num, ids, feats, edges, cet, efeats = load_preprocessed_typed_graph(cfg)
        dags = sampling_op_dags(cfg, anchor_target)
        rn_prefixes = cfg.random_negative_tfrecord_uri_prefixes
        out: Dict[str, List[str]] = {}
        s = HipGraphDBSampler(node_types, num, edges, cet, feats, device=device, sampling_seed=seed, edge_features=efeats)
        try:
            for t in anchor_target:
                if t not in rn_prefixes:
                    raise ValueError(f"nodeTypeToRandomNegativeTfrecordUriPrefix is missing node type {t!r}")
                w = _PartWriter(_res(cfg, rn_prefixes[t]))
                for i in range(0, ids[t].size, batch_size):
                    buf, off = s.encode_records_device(ids[t][i:i + batch_size], t, dags[t], tfrecord_frame=True)
                    w.add(_frames_to_host(buf), off.cpu().numpy())
                out[f"random_negative/{t}"] = w.close()
            roots = ids[pos_et.src_node_type]
            cap = cfg.num_max_training_samples_to_output
            if 0 < cap < roots.size:  # (the reference samples a fraction; a seeded choice of exactly `cap` roots here)
                roots = np.sort(np.random.default_rng(seed).choice(roots, size=cap, replace=False))
            w = _PartWriter(_res(cfg, cfg.nablp_tfrecord_uri_prefix))
            keep_isolated = cfg.should_include_isolated_nodes_in_training
            for i in range(0, roots.size, batch_size):
                buf, off, n_pos = s.encode_nablp_records_device(roots[i:i + batch_size], pos_et, cfg.num_positive_samples,
                                                                dags[pos_et.src_node_type], dags[pos_et.dst_node_type])
                w.add(_frames_to_host(buf), off.cpu().numpy(), None if keep_isolated else (n_pos > 0).cpu().numpy())
            out["node_anchor_based_link_prediction"] = w.close()
        finally:
            s.close()
        return out

    # Sampling, per-root assembly (createSubgraph), hydration, proto encoding and TFRecord framing all run on the
    # device (gigl_sample_khop + gigl_records_encode); the host only copies finished frames into the part files.
    @staticmethod
    def _run_node_classification(cfg, svc: HipKHopSamplerService, ids: np.ndarray, labels, batch_size: int):
        """createRootedNodeNeighborhoodSubgraph (SGSPureSparkV1Task.scala:973-1017): one RootedNodeNeighborhood per
        node; createSupervisedNodeClassificationSubgraph (SupervisedNodeClassificationTask.scala:166-236): labeled
        samples for the roots that have a label and at least one edge"""
        import torch
        eng = svc.engine
        pm = cfg.preprocessed_metadata.nodes[0]
        unl = _PartWriter(cfg.unlabeled_tfrecord_uri_prefix)
        lab = _PartWriter(cfg.labeled_tfrecord_uri_prefix)
        # numMaxTrainingSamplesToOutput: the reference keeps an arbitrary `LIMIT n` of the training samples
        # (downsampleNumberOfNodes, SGSPureSparkV1Task.scala:1042-1081); here: the first n in node-id order
        limit = cfg.num_max_training_samples_to_output
        # shouldSkipTraining && shouldSkipModelEvaluation: no labeled samples (SupervisedNodeClassificationTask.scala:101-106)
        skip_labeled = cfg.should_skip_training and cfg.should_skip_model_evaluation
        for i in range(0, ids.size, batch_size):
            chunk = ids[i:i + batch_size]
            tree = eng.sample_khop(chunk, cfg.fanouts, sampling_seed=svc.sampling_seed, mode=getattr(svc, 'sampling_mode', 0))
            buf, off = eng.encode_records(tree)
            unl.add(_frames_to_host(buf), off.cpu().numpy())
            if skip_labeled:
                continue
            sfx, sfx_off = _encode_labels(pm.label_keys, labels, chunk)
            has_label = torch.from_numpy(np.diff(sfx_off) > 0).to(eng.device)
            emit = has_label & (tree.cnt[0] > 0)  # isolated nodes produce no training samples
            if limit > 0:
                emit = emit & ((torch.cumsum(emit.to(torch.int64), 0) + lab.n_records) <= limit)
            emit = emit.to(torch.uint8)
            buf, off = eng.encode_records(tree, emit=emit, suffix=torch.from_numpy(sfx), suffix_off=torch.from_numpy(sfx_off))
            lab.add(_frames_to_host(buf), off.cpu().numpy())
        return {"unlabeled": unl.close(), "labeled": lab.close()}

    @staticmethod
    def _run_nablp(cfg, svc: HipKHopSamplerService, ids: np.ndarray, batch_size: int):
        """createNodeAnchorBasedLinkPredictionSubgraph (NodeAnchorBasedLinkPredictionTask.scala:146-312):
        neighborhood = array_distinct(root nbhd ++ union of the positives' nbhds) — the positives' rooted samples are
        re-derived on the device (the sample of a root is a pure function of the root and the seed, so this equals
        the reference's lookup in its cached per-node table); pos_edges = [root -> pos]; hard_neg_edges = [] unless the
        preprocessed metadata names user-defined label edges (see below); neg_edges = []"""
        import torch
        from . import _lib
        eng = svc.engine
        # User-defined labels (UserDefinedLabelsNodeAnchorBasedLinkPredictionTask.scala): positives / hard negatives are
        # sampled from the user's own edge lists (counter 3 / 4), their neighbourhoods are merged into the sample and
        # the label edges carry the user tables' features; roots = nodes with at least one user-defined positive.
        em = cfg.preprocessed_metadata.edges[0]
        pos_ud, neg_ud = em.positive_edge_info is not None, em.negative_edge_info is not None
        P, Q = cfg.num_positive_samples, 0
        n_ids = eng.n_nodes
        if pos_ud:
            P = cfg.num_user_defined_positive_samples
            assert P > 0, ("numUserDefinedPositiveSamples must be provided in subgraphSamplerConfig and > 0 if user "
                           "defined pos edges are provided")
            eng.load_label_edges("pos", n_ids, *load_label_edge_table(cfg, em.positive_edge_info))
        if neg_ud:
            Q = cfg.num_user_defined_negative_samples
            assert Q > 0, ("numUserDefinedNegativeSamples must be provided in subgraphSamplerConfig and > 0 if user "
                           "defined neg edges are provided")
            eng.load_label_edges("neg", n_ids, *load_label_edge_table(cfg, em.negative_edge_info))
        T = 1 + P + Q
        main = _PartWriter(cfg.nablp_tfrecord_uri_prefix)
        rn = {t: _PartWriter(p) for t, p in cfg.random_negative_tfrecord_uri_prefixes.items()}
        limit = cfg.num_max_training_samples_to_output  # LIMIT n of the main samples (first n in node-id order)
        # shouldSkipTraining && shouldSkipModelEvaluation: only the RootedNodeNeighborhood samples are written
        # (NodeAnchorBasedLinkPredictionTask.scala:111-116)
        skip_main = cfg.should_skip_training and cfg.should_skip_model_evaluation
        for i in range(0, ids.size, max(1, batch_size // T)):
            chunk = ids[i:i + max(1, batch_size // T)]
            roots = eng._roots_tensor(chunk)
            if skip_main:
                self_write_rn(eng, svc, cfg, roots, rn)
                continue
            pos, cnt = eng.sample_positives(roots, P, sampling_seed=svc.sampling_seed,
                                            label_edges="pos" if pos_ud else None)
            cols = [roots.view(-1, 1), pos.view(-1, P)]
            if neg_ud:
                neg, _ = eng.sample_positives(roots, Q, sampling_seed=svc.sampling_seed, counter=4, label_edges="neg")
                cols.append(neg.view(-1, Q))
            grouped = torch.cat(cols, dim=1).reshape(-1).contiguous()
            tree = eng.sample_khop(grouped, cfg.fanouts, sampling_seed=svc.sampling_seed, mode=getattr(svc, 'sampling_mode', 0))
            # anchors need at least one positive and, in the task without user-defined labels, a neighbourhood of their
            # own: the reference INNER JOINs the sampled positives with subgraphVIEW, which holds the roots with at least
            # one in-edge (NodeAnchorBasedLinkPredictionTask.scala:186-194; createSubgraph "does not include isolated
            # nodes", :69).  The user-defined-labels task joins against the RootedNodeNeighborhood view instead, which
            # has the neighbourless nodes too (UserDefinedLabelsNodeAnchorBasedLinkPredictionTask.scala:365-384).
            emit = cnt > 0
            if not (pos_ud or neg_ud):
                emit = emit & (tree.cnt[0].view(-1, T)[:, 0] > 0)
            if limit > 0:
                emit = emit & ((torch.cumsum(emit.to(torch.int64), 0) + main.n_records) <= limit)
            buf, off = eng.encode_records(tree, kind=_lib.REC_NODE_ANCHOR_LINK_PRED, trees_per_record=T,
                                          emit=emit.to(torch.uint8), n_neg_trees=Q,
                                          pos_label_edges="pos" if pos_ud else None,
                                          neg_label_edges="neg" if neg_ud else None)
            main.add(_frames_to_host(buf), off.cpu().numpy())
            self_write_rn(eng, svc, cfg, roots, rn)
        files = {"node_anchor_based_link_prediction": main.close()}
        for t, w in rn.items():
            files[f"random_negative/{t}"] = w.close()
        return files


def load_preprocessed_typed_graph(cfg: GbmlConfigPbWrapper):
    """loadHydratedNodeDataFrame / loadHydratedEdgeDataFrame for every condensed type (scala_spark35 SGSTask.scala;
    GraphDBNodeAnchorBasedLinkPredictionTask.scala:156-166): per node type its ids and feature rows, per edge type its
    (src, dst) list and edge feature rows, decoded by the native reader"""
    from .graphdb_sampler import EdgeType
    from .ingest import COL_F32, COL_I64, feature_widths, read_columns
    pm = cfg.preprocessed_metadata
    node_types = {name: c for c, name in cfg.condensed_node_type_map.items()}
    ids: Dict[str, np.ndarray] = {}
    feats: Dict[str, np.ndarray] = {}
    num: Dict[str, int] = {}
    for c, name in cfg.condensed_node_type_map.items():
        nm = pm.nodes[c]
        files = tfrecord_files(os.path.join(_res(cfg, nm.tfrecord_uri_prefix), ""))
        widths = feature_widths(files[0], nm.feature_keys) if files and nm.feature_keys else []
        cols = [(nm.node_id_key, COL_I64, 1)] + [(k, COL_F32, max(w, 1)) for k, w in zip(nm.feature_keys, widths)]
        data, _ = read_columns(files, cols)
        nid = data[nm.node_id_key][:, 0]
        ids[name] = np.unique(nid)
        num[name] = int(nid.max()) + 1 if nid.size else 0
        if sum(widths):
            x = np.zeros((num[name], int(sum(widths))), dtype=np.float32)
            x[nid] = np.concatenate([data[k][:, :w] for k, w in zip(nm.feature_keys, widths) if w], axis=1)
            feats[name] = x
    edges, efeats, cet = {}, {}, {}
    for c, (s_t, rel, d_t) in cfg.condensed_edge_type_map.items():
        em = pm.edges[c]
        et = EdgeType(s_t, rel, d_t)
        files = tfrecord_files(os.path.join(_res(cfg, em.tfrecord_uri_prefix), ""))
        widths = feature_widths(files[0], em.feature_keys) if files and em.feature_keys else []
        cols = [(em.src_node_id_key, COL_I64, 1), (em.dst_node_id_key, COL_I64, 1)]
        cols += [(k, COL_F32, max(w, 1)) for k, w in zip(em.feature_keys, widths)]
        ed, _ = read_columns(files, cols)
        edges[et] = (ed[em.src_node_id_key][:, 0].astype(np.uint32), ed[em.dst_node_id_key][:, 0].astype(np.uint32))
        cet[et] = c
        if sum(widths):
            efeats[et] = np.concatenate([ed[k][:, :w] for k, w in zip(em.feature_keys, widths) if w], axis=1)
    return node_types, num, ids, feats, edges, cet, efeats


def default_sampling_op_dag(cfg: GbmlConfigPbWrapper, root_node_type: str):
    """the k-hop message-passing DAG of a root node type when the config names no SubgraphSamplingStrategy: hop 1 =
    one INCOMING op per edge type that ends in the root type, hop h+1 = for every hop-h op one INCOMING op per edge
    type that ends in that op's source type; numNeighborsToSample neighbours each, numHops hops (the uniform k-hop
    sampling the homogeneous sampler does, per edge type)"""
    from .graphdb_sampler import INCOMING, EdgeType, SamplingOp, SamplingOpDAG
    ets = [EdgeType(*t) for _, t in sorted(cfg.condensed_edge_type_map.items())]
    f = cfg.num_neighbors_to_sample
    ops, level = [], [(None, root_node_type)]
    for hop in range(1, cfg.num_hops + 1):
        nxt = []
        for parent, ntype in level:
            for et in ets:
                if et.dst_node_type != ntype:
                    continue
                name = f"hop{hop}_{len(ops)}_{et.relation}"
                ops.append(SamplingOp(name, et, f, [parent] if parent else [], INCOMING))
                nxt.append((name, et.src_node_type))
        level = nxt
    return SamplingOpDAG.from_ops(ops)


def sampling_op_dags(cfg: GbmlConfigPbWrapper, node_types: Sequence[str]):
    """getNodeTypeToSamplingOpDagMap (SubgraphSamplingStrategyWrapper.scala:10-20) for the node types asked for"""
    from .graphdb_sampler import (EdgeType, SamplingOp, SamplingOpDAG, SubgraphSamplingValidationError,
                                  validate_sampling_op_dags)
    out = {}
    raw: Dict[str, list] = {}
    for path in cfg.message_passing_paths:
        if str(path["rootNodeType"]) in raw:
            raise SubgraphSamplingValidationError("REPEATED_ROOT_NODE_TYPE", f"two DAGs for {path['rootNodeType']!r}")
        ops = []
        for op in path.get("samplingOps") or []:
            et = op["edgeType"]
            ops.append(SamplingOp(
                op["opName"], EdgeType(et["srcNodeType"], et["relation"], et["dstNodeType"]),
                int((op.get("randomUniform") or {}).get("numNodesToSample", 0) or op.get("numNodesToSample", 0)),
                list(op.get("inputOpNames") or []), str(op.get("samplingDirection", "INCOMING"))))
        raw[str(path["rootNodeType"])] = ops
    validate_sampling_op_dags(raw, list(cfg.condensed_node_type_map.values()),
                              [EdgeType(*t) for t in cfg.condensed_edge_type_map.values()])
    for root_type, ops in raw.items():
        out[root_type] = SamplingOpDAG.from_ops(ops)
    for t in node_types:
        if t not in out:
            out[t] = default_sampling_op_dag(cfg, t)
    return out


def self_write_rn(eng, svc, cfg, roots, rn) -> None:
    """the RootedNodeNeighborhood sample of every node (random-negative stream / inference input) for one chunk"""
    if not rn:
        return
    tree = eng.sample_khop(roots, cfg.fanouts, sampling_seed=svc.sampling_seed, mode=getattr(svc, 'sampling_mode', 0))
    buf, off = eng.encode_records(tree)
    b_h, o_h = _frames_to_host(buf), off.cpu().numpy()
    for w in rn.values():
        w.add(b_h, o_h)


def main(argv=None):
    ap = argparse.ArgumentParser(description="MI355X subgraph sampler (drop-in for gigl.src.subgraph_sampler)")
    ap.add_argument("--job_name", required=True)
    ap.add_argument("--task_config_uri", required=True)
    ap.add_argument("--resource_config_uri", default=None)
    ap.add_argument("--uri_base", default=None, help="base directory for relative URIs in the configs")
    a = ap.parse_args(argv)
    files = SubgraphSampler().run(a.job_name, a.task_config_uri, a.resource_config_uri, uri_base=a.uri_base)
    for k, v in files.items():
        print(k, len(v), "file(s)")


if __name__ == "__main__":
    main()
