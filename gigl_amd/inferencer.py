"""Inferencer — drop-in for the reference component

    python -m gigl.src.inference.inferencer --job_name --task_config_uri --resource_config_uri
    Inferencer().run(applied_task_identifier, task_config_uri, resource_config_uri, custom_worker_image_uri=None,
                     cpu_docker_uri=None, cuda_docker_uri=None)          (python/gigl/src/inference/inferencer.py:25-33)

Reference body (python/gigl/src/inference/v1/gnn_inferencer.py:113-140,234-340; v1/lib/utils.py:78-228;
v1/lib/base_inference_blueprint.py:52-103): build the plugin, load the trained state_dict, read the
RootedNodeNeighborhood (or SupervisedNodeClassificationSample) TFRecords, batch `inference_batch_size`
(default 3000) samples, `infer_batch`, and emit ONE row per ROOT in batch order — {"node_id", "emb": [...]}
and/or {"node_id", "pred"} — as JSON lines (the Beam/Dataflow + BigQuery load is cloud plumbing, out of scope:
rows are written to the local paths named in inferenceMetadata).
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict, Optional

import numpy as np
import torch

from .base import BaseInferencer, import_obj
from .batches import RootedNodeNeighborhoodBatch, SupervisedNodeClassificationBatch, iterate_tfrecord_batches
from .config import GbmlConfigPbWrapper, _get, resolve_uri, tfrecord_files
from . import wire


def generate_inferencer_instance(cfg: GbmlConfigPbWrapper) -> BaseInferencer:
    cls_path = cfg.inferencer_cls_path
    if not cls_path:
        raise ValueError("inferencerConfig.inferencerClsPath is not set")
    inferencer = import_obj(cls_path)(**cfg.inferencer_args)
    assert isinstance(inferencer, BaseInferencer)
    state_dict = torch.load(cfg.trained_model_uri, map_location="cpu")
    inferencer.init_model(gbml_config_pb_wrapper=cfg, state_dict=state_dict)
    return inferencer


class _RowWriter:
    """one output row per root, in batch order: {"node_id", "emb"} / {"node_id", "pred"}
    (base_inference_blueprint.py:76-103).  An embeddingsPath that ends in "/" is a directory of Avro shards written by
    the device-side EmbeddingExporter (the format the reference's exporter hands to BigQuery,
    python/gigl/common/data/export.py); a file path gets line-per-root JSON, formatted natively
    (gigl_json_rows_format).  WORLD_SIZE > 1: every rank writes its own files (Avro shards prefixed rank_<r>, JSON
    files suffixed .rank<r>)."""

    def __init__(self, out_files: Dict[str, str], node_type: str, rank: int = 0, world: int = 1,
                 keep_on_device: bool = False):
        self.node_type, self.n_rows = node_type, 0
        self.exporter = self.emb_fh = self.pred_fh = None
        sfx = f".rank{rank}" if world > 1 else ""
        for p in out_files.values():
            os.makedirs(os.path.dirname(p.rstrip("/")) or ".", exist_ok=True)
        emb = out_files.get("embeddings")
        if emb and emb.endswith("/"):
            from .export import EmbeddingExporter
            self.exporter = EmbeddingExporter(emb, file_prefix=(f"rank_{rank}" if world > 1 else None),
                                              min_shard_size_threshold_bytes=1 << 28, keep_on_device=keep_on_device)
        elif emb:
            out_files["embeddings"] = emb + sfx
            self.emb_fh = open(emb + sfx, "wb")
        if out_files.get("predictions"):
            out_files["predictions"] += sfx
            self.pred_fh = open(out_files["predictions"], "wb")

    @staticmethod
    def _json(ids: np.ndarray, emb: Optional[torch.Tensor], pred: Optional[torch.Tensor]) -> memoryview:
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        n = int(ids.size)
        e = None if emb is None else np.ascontiguousarray(emb.detach().to("cpu", torch.float32).numpy())
        pr = None if pred is None else np.ascontiguousarray(pred.detach().to("cpu", torch.int32).numpy())
        d = 0 if e is None else int(e.shape[1])
        cap = int(lib.gigl_json_rows_capacity(n, d))
        buf = np.empty(cap, dtype=np.uint8)
        used = C.c_int64()
        _lib.check(lib.gigl_json_rows_format(
            C.c_void_p(ids.ctypes.data), C.c_void_p(e.ctypes.data) if e is not None else None, d,
            C.c_void_p(pr.ctypes.data) if pr is not None else None, n, d, C.c_void_p(buf.ctypes.data), cap,
            C.byref(used)))
        return memoryview(buf[: used.value])

    def add(self, ids: np.ndarray, embeddings: Optional[torch.Tensor], predictions: Optional[torch.Tensor],
            ids_dev: Optional[torch.Tensor] = None) -> None:
        """ids: the rows' node ids on the host; ids_dev: the same on the device when the caller has them there (the
        device-side encoder then needs no upload)"""
        if self.exporter is not None and embeddings is not None:
            self.exporter.add_embedding(ids_dev if ids_dev is not None else torch.from_numpy(np.asarray(ids, dtype=np.int64)),
                                        embeddings, self.node_type)
        if self.emb_fh is not None and embeddings is not None:
            self.emb_fh.write(self._json(ids, embeddings, None))
        if self.pred_fh is not None and predictions is not None:
            self.pred_fh.write(self._json(ids, None, predictions))
        self.n_rows += int(np.asarray(ids).size)

    def close(self) -> None:
        if self.exporter is not None:
            self.exporter.close()
        for fh in (self.emb_fh, self.pred_fh):
            if fh is not None:
                fh.close()


class Inferencer:
    def run(self, applied_task_identifier: str, task_config_uri: str, resource_config_uri: Optional[str] = None,
            custom_worker_image_uri: Optional[str] = None, cpu_docker_uri: Optional[str] = None,
            cuda_docker_uri: Optional[str] = None, *, uri_base: Optional[str] = None, device: Optional[int] = None,
            route: Optional[str] = None) -> Dict[str, str]:
        """route: "hbm" (graph + features resident in HBM, batches sampled there: gigl_amd/hbm.py), "tfrecord" (the
        sampler's RootedNodeNeighborhood files, the reference's dataflow) or None = the plugin's `data_route` argument /
        GIGL_AMD_ROUTE / auto.  Both routes walk the same roots in the same batches and write the same rows.
        One process per GPU: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment, as the trainer."""
        if not torch.cuda.is_available():
            raise RuntimeError("gigl_amd.Inferencer needs a HIP device; there is no CPU fallback")
        import torch.distributed as dist
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        dev_index = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        started_pg = False
        if world > 1 and not dist.is_initialized():
            dist.init_process_group(backend=os.environ.get("GIGL_DIST_BACKEND") or "nccl")
            started_pg = True
        try:
            cfg = GbmlConfigPbWrapper.from_uri(task_config_uri, uri_base=uri_base)
            inferencer = generate_inferencer_instance(cfg)
            inferencer.model = inferencer.model.to(dev)
            if cfg.is_heterogeneous:
                # typed graphs: "hbm" = the typed tables resident in HBM, every batch's typed graph built there by the
                # one-call plan (gigl_typed_plan_*); auto / "tfrecord" = the sampler's typed RootedNodeNeighborhood files
                want = (route or cfg.inferencer_args.get("data_route") or os.environ.get("GIGL_AMD_ROUTE") or "auto").lower()
                if want not in ("hbm", "tfrecord", "auto"):
                    raise ValueError(f"data route {want!r}: expected hbm, tfrecord or auto")
                self.route = "hbm" if want == "hbm" else "tfrecord"
                if self.route == "hbm":
                    # (WORLD_SIZE > 1: every rank holds the typed tables and takes the batches c % world == rank)
                    return self._run_typed_hbm(cfg, inferencer, dev, rank, world)
                return self._run_typed(cfg, inferencer, dev)
            from .hbm import route_of
            self.route = route_of(cfg, cfg.inferencer_args, route)
            if self.route == "hbm" and not getattr(inferencer, "supports_hbm_batches", False):
                if route == "hbm":
                    raise NotImplementedError(f"{type(inferencer).__name__} does not take in-HBM batches")
                self.route = "tfrecord"
            info = (_get(cfg.doc, "sharedConfig.inferenceMetadata.nodeTypeToInferencerOutputInfoMap", {}) or {})
            out_files: Dict[str, str] = {}
            for _, v in info.items():
                if v.get("embeddingsPath"):
                    out_files["embeddings"] = resolve_uri(v["embeddingsPath"], cfg.uri_base)
                if v.get("predictionsPath"):
                    out_files["predictions"] = resolve_uri(v["predictionsPath"], cfg.uri_base)
            writer = _RowWriter(out_files, str(cfg.node_types[0]), rank, world)
            try:
                if self.route == "hbm":
                    self._run_hbm(cfg, inferencer, dev, writer, rank, world)
                else:
                    if world > 1:
                        raise NotImplementedError("the TFRecord route runs in one process; WORLD_SIZE > 1 takes the "
                                                  "in-HBM route (route='hbm')")
                    self._run_tfrecord(cfg, inferencer, dev, writer)
            finally:
                writer.close()
            self.rows_written = writer.n_rows
            return out_files
        finally:
            if started_pg:
                dist.destroy_process_group()

    @staticmethod
    def _run_tfrecord(cfg: GbmlConfigPbWrapper, inferencer, dev, writer: _RowWriter) -> None:
        """the reference's dataflow: RootedNodeNeighborhood TFRecords -> collate -> infer_batch"""
        if cfg.task_kind == "node_classification":
            files = tfrecord_files(cfg.unlabeled_tfrecord_uri_prefix)
        else:
            files = [f for p in cfg.random_negative_tfrecord_uri_prefixes.values() for f in tfrecord_files(p)]
        for raw in iterate_tfrecord_batches(files, cfg.inference_batch_size):
            rnn = RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(raw, cfg.node_types[0])
            if cfg.task_kind == "node_classification":
                batch = SupervisedNodeClassificationBatch(
                    graph=rnn.graph, root_node_indices=rnn.condensed_node_type_to_root_node_indices_map[0],
                    root_nodes=rnn.root_nodes, root_node_labels=None)
            else:  # link prediction plugins take the RootedNodeNeighborhoodBatch (utils.py:78-228)
                batch = rnn
            res = inferencer.infer_batch(batch=batch, device=dev)
            writer.add(np.array([r.id for r in batch.root_nodes], dtype=np.int64), res.embeddings, res.predictions)

    def _run_hbm(self, cfg: GbmlConfigPbWrapper, inferencer, dev, writer: _RowWriter, rank: int, world: int) -> None:
        """graph + features resident in HBM (this rank's shard at WORLD_SIZE > 1), every batch sampled, union-ed and
        encoded there by the one-call plan; the same roots in the same batches as _run_tfrecord"""
        from .hbm import ResidentGraph
        resident = ResidentGraph(cfg, dev, rank=rank, world=world)
        try:
            self.infer_resident(inferencer, dev, resident, writer, cfg.inference_batch_size)
        finally:
            resident.close()

    def infer_resident(self, inferencer, dev, resident, writer: _RowWriter, batch_size: int,
                       groups: Optional[int] = None) -> None:
        """the in-HBM inference pass over every node of `resident`: batches of `batch_size` roots in the TFRecord
        route's order, `groups` batches per library call, rows handed to `writer` (device tensors: the Avro writer
        encodes them on the device and writes them out on its own thread while the next call computes)"""
        ids = resident.inference_root_order()
        b = int(batch_size)
        slots = 1
        for f in reversed(resident.fanouts):
            slots = 1 + f * slots
        n_batches = -(-ids.size // b) if ids.size else 0
        per_rank = max(1, -(-n_batches // resident.world))
        if groups is None:  # batches per library call: as many as keep the call's tree under ~2^26 slots, at most 64
            groups = int(max(1, min(64, per_rank, (1 << 26) // max(b * slots, 1))))
        self.hbm_groups = groups
        # Calls of the one-call plan rotate over a few lanes — a ctx, plan set and stream each, the resident graph shared
        # (ResidentGraph.lane_engine) — so that one call's sampler / union run under another's layers, as in bench.py;
        # rows are handed to the writer on ONE stream, in batch order.  (GIGL_AMD_INFER_STREAMS=1: one stream.)
        n_lanes = max(1, int(os.environ.get("GIGL_AMD_INFER_STREAMS", "3")))
        if getattr(resident, "sharded", False) or not hasattr(resident, "lane_engine"):
            n_lanes = 1
        main = torch.cuda.current_stream(dev)
        lanes = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)] if n_lanes > 1 else []
        wstream = torch.cuda.Stream(device=dev) if n_lanes > 1 else None  # the writer's: rows arrive in batch order
        use_lanes = False
        # Calls on the lanes are PIPELINED: call c's rows are handed to the writer when call c + 2 n_lanes has been issued.  By
        # then call c is long finished, so settling it (did its batch outgrow the plan's workspace? then its rows are NaN and
        # the batch is encoded again through the staged launches, on its lane) costs no stall — and no call's rows reach the
        # writer unchecked, whatever the graph's size.
        pending = []
        depth = 2 * n_lanes  # (the oldest unsettled call finished long ago: settling it never drains the launch queue)

        def hand_over(entry):
            hb_, res_, st_, done = entry
            if getattr(resident, "call_overflowed", lambda _b: False)(hb_):
                hb_.force_staged, hb_.defer_overflow_check = True, False
                with torch.cuda.stream(st_):
                    res_ = inferencer.infer_batch(batch=hb_, device=dev)
                    done = torch.cuda.Event()
                    done.record(st_)
            wstream.wait_event(done)  # (the call's own end: its lane has later calls queued behind it by now)
            with torch.cuda.stream(wstream):
                for t in (res_.embeddings, res_.predictions):
                    if t is not None:
                        t.record_stream(wstream)
                if hb_.root_ids.size:
                    writer.add(hb_.root_ids, res_.embeddings, res_.predictions, ids_dev=hb_.root_ids_dev)
        for c, hb in enumerate(resident.root_batches(ids, b, groups)):
            if not use_lanes:
                res = inferencer.infer_batch(batch=hb, device=dev)  # (checked and, if need be, redone inside encode)
                if hb.root_ids.size:
                    writer.add(hb.root_ids, res.embeddings, res.predictions, ids_dev=hb.root_ids_dev)
                if c == 0 and n_lanes > 1 and any(p is not None for p in getattr(resident, "_plans", {}).values()):
                    # the batch went through a one-call plan: from here on a lane (ctx + plan set + stream) per call.
                    # Everything so far — the roots' upload, the first rows — is on the main stream: the lanes and the
                    # writer's stream start behind it
                    use_lanes = True
                    for st in lanes + [wstream]:
                        st.wait_stream(main)
                continue
            lane = 1 + (c % n_lanes)  # (lane 0 is the main stream's ctx: left to the first batch)
            hb.lane = lane
            hb.defer_overflow_check = True
            with torch.cuda.stream(lanes[lane - 1]):
                res = inferencer.infer_batch(batch=hb, device=dev)
                done = torch.cuda.Event()
                done.record(lanes[lane - 1])
            pending.append((hb, res, lanes[lane - 1], done))
            if len(pending) >= depth:
                hand_over(pending.pop(0))
        while pending:
            hand_over(pending.pop(0))
        if use_lanes:
            main.wait_stream(wstream)
            for st in lanes:
                main.wait_stream(st)
        if hasattr(resident, "raise_on_overflow"):
            resident.raise_on_overflow()  # (every pipelined call was settled above: nothing to report)
        self.hbm_overflow_redone = int(getattr(resident, "overflow_redone", 0))  # calls redone through the staged launches


def _typed_run(self, cfg: GbmlConfigPbWrapper, inferencer, dev) -> Dict[str, str]:
    """heterogeneous jobs: every node type named in nodeTypeToInferencerOutputInfoMap gets the embeddings of its own
    RootedNodeNeighborhood samples (typed collate -> the encoder's rows of that type), one JSON line per root"""
    from .batches import HeteroRootedNodeNeighborhoodBatch
    info = (_get(cfg.doc, "sharedConfig.inferenceMetadata.nodeTypeToInferencerOutputInfoMap", {}) or {})
    prefixes = cfg.random_negative_tfrecord_uri_prefixes
    out_files: Dict[str, str] = {}
    n_rows = 0
    for node_type, v in info.items():
        if not v.get("embeddingsPath") or node_type not in prefixes:
            continue
        files = {"embeddings": resolve_uri(v["embeddingsPath"], cfg.uri_base)}
        # one output per node type (the reference writes a table per type: rows {"node_id", "emb"}), through the writer
        # of the homogeneous routes: JSON lines formatted natively, or Avro shards when the path names a directory
        writer = _RowWriter(files, node_type)
        try:
            for raw in iterate_tfrecord_batches(tfrecord_files(prefixes[node_type]), cfg.inference_batch_size):
                batch = HeteroRootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(
                    raw, cfg.condensed_node_type_map, cfg.condensed_edge_type_map)
                emb = inferencer.infer_batch(batch=batch, device=dev).embeddings
                writer.add(np.array([gid for _, gid in batch.root_nodes], dtype=np.int64), emb.float(), None)
        finally:
            writer.close()
        out_files[f"embeddings/{node_type}"] = files["embeddings"]
        n_rows += writer.n_rows
    self.rows_written = n_rows
    return out_files


Inferencer._run_typed = _typed_run


def _typed_run_hbm(self, cfg: GbmlConfigPbWrapper, inferencer, dev, rank: int = 0, world: int = 1) -> Dict[str, str]:
    """heterogeneous jobs, in-HBM route: the preprocessor's typed tables are read once into HBM (one CSR per edge type
    and direction, one feature table per node type), every batch's typed graph — the config's SamplingOp DAG for the
    roots, the distinct nodes per type, the distinct edges per edge type — is built there by the library's one-call
    plan (HipGraphDBSampler.batch_graph_plan -> gigl_typed_plan_*), and the encoder runs over it.  The same roots in the
    same batches as the TFRecord route (the sampler writes a type's roots in table order); the rows differ from that
    route's by fp32 summation order only (a type's nodes are numbered ascending here, first-seen there).
    WORLD_SIZE > 1: a replica of the typed tables per rank (they fit one GPU wherever this route is used), rank r takes
    the batches c with c % world == r and writes its own files (suffix .rank<r> / shard prefix rank_<r>), like the
    homogeneous route's file sharding (data_loaders/utils.py:23-56)."""
    from .graphdb_sampler import HipGraphDBSampler
    from .hbm import planned_root_order
    from .subgraph_sampler import load_preprocessed_typed_graph, sampling_op_dags
    info = (_get(cfg.doc, "sharedConfig.inferenceMetadata.nodeTypeToInferencerOutputInfoMap", {}) or {})
    wanted = [t for t, v in info.items() if v.get("embeddingsPath")]
    node_types, num, ids, feats, edges, cet, efeats = load_preprocessed_typed_graph(cfg)
    inner = inferencer.model.module if hasattr(inferencer.model, "module") else inferencer.model
    enc = getattr(inner, "_encoder", getattr(inner, "encoder", inner))
    is_hgt = type(enc).__name__ == "HGT"  # (HGT takes the row subset; the link-prediction wrapper passes it through)
    dags = sampling_op_dags(cfg, wanted)
    # the sampler job's rule (SubgraphSampler.run): "deterministic" = the hash permutation under seed 42 — the samples of
    # its files, so both routes see the same neighbourhoods; anything else = a fresh uniform sample per job
    seed = 42 if cfg.permutation_strategy == "deterministic" else 1 + int.from_bytes(os.urandom(3), "little") % ((1 << 20) - 1)
    # (edge feature rows ride along for the encoders that read them — SimpleHGN's attention: the batch graph's
    # edge_attr_dict, joined on the device; HGT ignores them)
    has_ef = (not is_hgt) and any(np.asarray(v).size for v in (efeats or {}).values())
    s = HipGraphDBSampler(node_types, num, edges, cet, feats, device=dev.index or 0, sampling_seed=seed,
                          edge_features=efeats if has_ef else None)
    out_files: Dict[str, str] = {}
    n_rows = 0
    b = int(cfg.inference_batch_size)
    try:
        if hasattr(inferencer, "_ensure_engine"):
            inferencer._ensure_engine(dev)
        for node_type in wanted:
            path = resolve_uri(info[node_type]["embeddingsPath"], cfg.uri_base)
            files = {"embeddings": path}
            # rows through the same writer as the homogeneous routes: line-per-root JSON formatted natively
            # (gigl_json_rows_format), or Avro shards encoded on the device when the path names a directory
            writer = _RowWriter(files, node_type, rank, world)
            try:
                prefix = cfg.random_negative_tfrecord_uri_prefixes.get(node_type)
                order = planned_root_order(np.asarray(ids[node_type]), prefix) if prefix else np.asarray(ids[node_type])
                # (HGT: the plan also lays out the layers' merged CSR by destination and the roots' rows of it)
                et_ids = enc.convs[0].edge_types_map if is_hgt and len(enc.convs) else None
                chunks = [np.asarray(order[i:i + b], dtype=np.int64) for i in range(0, order.size, b)][rank::world]
                # HGT encoders: the whole step — DAG sampler, typed batch graph, encoder, the roots' rows — is ONE library
                # call per batch, replayed as a hipGraph (csrc/hgt_plan.hip; GIGL_AMD_TYPED_ONE_CALL=0: staged launches)
                one_call = None
                if is_hgt and chunks and os.environ.get("GIGL_AMD_TYPED_ONE_CALL", "1") != "0":
                    from .models_hetero import HgtInferPlan
                    try:
                        one_call = HgtInferPlan(enc, s, node_type, dags[node_type], b)
                    except NotImplementedError:
                        one_call = None
                if one_call is not None:
                    try:
                        r_dev = [torch.from_numpy(c.astype(np.uint32).view(np.int32)).to(dev) for c in chunks]
                        torch.cuda.current_stream(dev).synchronize()  # (the uploads, before the plan's streams read them)
                        for ci, chunk in enumerate(chunks):
                            # (batch ci + 1 is announced: its graph part is built under this batch's layers)
                            emb = one_call.run(r_dev[ci], r_dev[ci + 1] if ci + 1 < len(chunks) else None)
                            torch.cuda.current_stream(dev).wait_stream(s.engine._stream)
                            writer.add(chunk, emb, None)
                        s.engine.synchronize()
                    finally:
                        one_call.close()
                    chunks = []
                issue = lambda c: s.batch_graph_plan_issue(c, node_type, dags[node_type], b_max=b, edge_type_ids=et_ids)
                ticket = issue(chunks[0]) if chunks else None
                for ci, chunk in enumerate(chunks):
                    # batch ci+1's sampling is enqueued before the model over batch ci is launched: the host never
                    # waits for a batch's counts with an idle device
                    graph, root_index, _ = s.batch_graph_plan_finish(ticket)
                    ticket = issue(chunks[ci + 1]) if ci + 1 < len(chunks) else None
                    with torch.no_grad():
                        if is_hgt:  # the last layer computes the roots' rows only
                            emb = inferencer.model(graph, [node_type], row_subset={node_type: root_index})[node_type]
                        else:
                            out = inferencer.model(graph, [node_type])[node_type]
                            emb = out[root_index.to(out.device)]
                    writer.add(chunk, emb.float(), None)
            finally:
                writer.close()
            out_files[f"embeddings/{node_type}"] = files["embeddings"]  # (the writer has put the rank suffix on it)
            n_rows += writer.n_rows
    finally:
        s.close()
    self.rows_written = n_rows
    return out_files


Inferencer._run_typed_hbm = _typed_run_hbm


def main(argv=None):
    ap = argparse.ArgumentParser(description="MI355X inferencer (drop-in for gigl.src.inference.inferencer)")
    ap.add_argument("--job_name", required=True)
    ap.add_argument("--task_config_uri", required=True)
    ap.add_argument("--resource_config_uri", default=None)
    ap.add_argument("--uri_base", default=None)
    a = ap.parse_args(argv)
    print(Inferencer().run(a.job_name, a.task_config_uri, a.resource_config_uri, uri_base=a.uri_base))


if __name__ == "__main__":
    main()
