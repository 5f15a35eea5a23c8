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


class Inferencer:
    def run(self, applied_task_identifier: str, task_config_uri: str, resource_config_uri: Optional[str] = None,
            custom_worker_image_uri: Optional[str] = None, cpu_docker_uri: Optional[str] = None,
            cuda_docker_uri: Optional[str] = None, *, uri_base: Optional[str] = None, device: int = 0) -> Dict[str, str]:
        if not torch.cuda.is_available():
            raise RuntimeError("gigl_amd.Inferencer needs a HIP device; there is no CPU fallback")
        cfg = GbmlConfigPbWrapper.from_uri(task_config_uri, uri_base=uri_base)
        dev = torch.device("cuda", device)
        inferencer = generate_inferencer_instance(cfg)
        inferencer.model = inferencer.model.to(dev)
        if cfg.is_heterogeneous:
            return self._run_typed(cfg, inferencer, dev)
        if cfg.task_kind == "node_classification":
            files = tfrecord_files(cfg.unlabeled_tfrecord_uri_prefix)
        else:
            files = [f for p in cfg.random_negative_tfrecord_uri_prefixes.values() for f in tfrecord_files(p)]
        info = (_get(cfg.doc, "sharedConfig.inferenceMetadata.nodeTypeToInferencerOutputInfoMap", {}) or {})
        out_files: Dict[str, str] = {}
        emb_fh = pred_fh = None
        for _, v in info.items():
            if v.get("embeddingsPath"):
                out_files["embeddings"] = resolve_uri(v["embeddingsPath"], cfg.uri_base)
            if v.get("predictionsPath"):
                out_files["predictions"] = resolve_uri(v["predictionsPath"], cfg.uri_base)
        for p in out_files.values():
            os.makedirs(os.path.dirname(p) or ".", exist_ok=True)
        # an embeddingsPath ending in "/" is a directory of Avro shards written by the device-side EmbeddingExporter
        # (the format the reference's exporter hands to BigQuery, python/gigl/common/data/export.py); a file path
        # keeps the line-per-root JSON output
        exporter = None
        if out_files.get("embeddings", "").endswith("/"):
            from .export import EmbeddingExporter
            exporter = EmbeddingExporter(out_files["embeddings"], min_shard_size_threshold_bytes=1 << 28)
        elif "embeddings" in out_files:
            emb_fh = open(out_files["embeddings"], "w")
        if "predictions" in out_files:
            pred_fh = open(out_files["predictions"], "w")
        n_rows = 0
        try:
            for raw in iterate_tfrecord_batches(files, cfg.inference_batch_size):
                rnn = RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(raw, cfg.node_types[0])
                if cfg.task_kind == "node_classification":
                    batch = SupervisedNodeClassificationBatch(
                        graph=rnn.graph, root_node_indices=rnn.condensed_node_type_to_root_node_indices_map[0],
                        root_nodes=rnn.root_nodes, root_node_labels=None)
                else:  # link prediction plugins take the RootedNodeNeighborhoodBatch (utils.py:78-228)
                    batch = rnn
                res = inferencer.infer_batch(batch=batch, device=dev)
                if exporter is not None and res.embeddings is not None:
                    ids = torch.tensor([r.id for r in batch.root_nodes], dtype=torch.int64)
                    exporter.add_embedding(ids, res.embeddings, str(cfg.node_types[0]))
                emb = res.embeddings.cpu() if res.embeddings is not None and emb_fh is not None else None
                pred = res.predictions.cpu() if res.predictions is not None else None
                for i, root in enumerate(batch.root_nodes):  # one row per root, in batch order
                    if emb_fh is not None and emb is not None:
                        emb_fh.write(json.dumps({"node_id": root.id, "emb": emb[i].tolist()}) + "\n")
                    if pred_fh is not None and pred is not None:
                        pred_fh.write(json.dumps({"node_id": root.id, "pred": int(pred[i])}) + "\n")
                    n_rows += 1
        finally:
            if exporter is not None:
                exporter.flush_embeddings()
            for fh in (emb_fh, pred_fh):
                if fh is not None:
                    fh.close()
        self.rows_written = n_rows
        return out_files


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
        path = resolve_uri(v["embeddingsPath"], cfg.uri_base)
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        out_files[f"embeddings/{node_type}"] = path
        with open(path, "w") as fh:
            for raw in iterate_tfrecord_batches(tfrecord_files(prefixes[node_type]), cfg.inference_batch_size):
                batch = HeteroRootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(
                    raw, cfg.condensed_node_type_map, cfg.condensed_edge_type_map)
                emb = inferencer.infer_batch(batch=batch, device=dev).embeddings.cpu()
                for i, (_, gid) in enumerate(batch.root_nodes):
                    fh.write(json.dumps({"node_id": int(gid), "node_type": node_type, "emb": emb[i].tolist()}) + "\n")
                    n_rows += 1
    self.rows_written = n_rows
    return out_files


Inferencer._run_typed = _typed_run


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
