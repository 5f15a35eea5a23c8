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
from .sampler_service import (HipKHopSamplerService, build_rooted_node_neighborhood, tree_to_edge_lists,
                              validate_rooted_node_neighborhood)


def load_preprocessed_graph(cfg: GbmlConfigPbWrapper):
    """loadNodeDataframeIntoSparkSql / loadEdgeDataframeIntoSparkSql (SGSPureSparkV1Task.scala:52-118,120-286):
    node ids are the dense enumerated ids of the Data Preprocessor; features = the featureKeys columns
    concatenated in order (:90-104)."""
    pm = cfg.preprocessed_metadata
    nm, em = pm.nodes[0], pm.edges[0]
    ids, feats, labels = [], [], {}
    for f in tfrecord_files(os.path.join(_res(cfg, nm.tfrecord_uri_prefix), "")):
        for rec in wire.read_tfrecords(f):
            ex = wire.decode_tf_example(rec)
            nid = int(np.asarray(ex[nm.node_id_key])[0])
            ids.append(nid)
            feats.append(np.concatenate([np.asarray(ex[k], dtype=np.float32).reshape(-1) for k in nm.feature_keys])
                         if nm.feature_keys else np.zeros(0, np.float32))
            for lk in nm.label_keys:
                if ex.get(lk) is not None and len(ex[lk]):
                    labels.setdefault(lk, {})[nid] = int(np.asarray(ex[lk])[0])
    n = (max(ids) + 1) if ids else 0
    d = feats[0].size if feats else 0
    x = np.zeros((n, d), dtype=np.float32)
    for nid, fv in zip(ids, feats):
        x[nid] = fv
    src, dst = [], []
    for f in tfrecord_files(os.path.join(_res(cfg, em.tfrecord_uri_prefix), "")):
        for rec in wire.read_tfrecords(f):
            ex = wire.decode_tf_example(rec)
            src.append(int(np.asarray(ex[em.src_node_id_key])[0]))
            dst.append(int(np.asarray(ex[em.dst_node_id_key])[0]))
    return n, np.asarray(src, dtype=np.uint32), np.asarray(dst, dtype=np.uint32), x, labels, sorted(set(ids))


def _res(cfg: GbmlConfigPbWrapper, uri: str) -> str:
    from .config import resolve_uri
    return resolve_uri(uri, cfg.uri_base)


def _write(prefix: str, payloads: List[bytes], records_per_file: int = 100_000) -> List[str]:
    """spark-tfrecord writes part files under the prefix directory (TFRecordIO.scala:53-69, overwrite mode)"""
    is_dir = prefix.endswith("/") or prefix.endswith(os.sep)
    d = prefix if is_dir else os.path.dirname(prefix)
    os.makedirs(d or ".", exist_ok=True)
    for old in tfrecord_files(prefix):
        os.remove(old)
    out = []
    for i in range(0, max(len(payloads), 1), records_per_file):
        name = (os.path.join(prefix, f"part-{i // records_per_file:05d}.tfrecord") if is_dir
                else f"{prefix}{i // records_per_file:05d}.tfrecord")
        wire.write_tfrecords(name, payloads[i:i + records_per_file])
        out.append(name)
    return out


class SubgraphSampler:
    def run(self, applied_task_identifier: str, task_config_uri: str, resource_config_uri: Optional[str] = None,
            cluster_name: Optional[str] = None, debug_cluster_owner_alias: Optional[str] = None,
            custom_worker_image_uri: Optional[str] = None, skip_cluster_delete: bool = False,
            additional_spark35_jar_file_uris: Sequence[str] = (), *, uri_base: Optional[str] = None,
            device: int = 0, batch_size: int = 4096) -> Dict[str, List[str]]:
        cfg = GbmlConfigPbWrapper.from_uri(task_config_uri, uri_base=uri_base)
        if cfg.permutation_strategy != "deterministic":
            raise NotImplementedError("only experimental_flags.permutation_strategy=deterministic is implemented "
                                      "(SamplingStrategy.scala:16-82); the non-deterministic F.shuffle has no parity")
        n, src, dst, x, labels, node_ids = load_preprocessed_graph(cfg)
        with HipKHopSamplerService(n, src, dst, x, cfg.is_graph_directed, device=device) as svc:
            rnns = self._sample_all(svc, node_ids, cfg.fanouts, batch_size)
            if cfg.task_kind == "node_classification":
                return self._run_node_classification(cfg, rnns, labels)
            return self._run_nablp(cfg, svc, rnns)

    # ---- shared: one RootedNodeNeighborhood per node (createRootedNodeNeighborhoodSubgraph)
    @staticmethod
    def _sample_all(svc: HipKHopSamplerService, node_ids: Sequence[int], fanouts: Sequence[int], batch_size: int):
        out: Dict[int, wire.RootedNodeNeighborhood] = {}
        for i in range(0, len(node_ids), batch_size):
            chunk = node_ids[i:i + batch_size]
            for nid, rnn in zip(chunk, svc.getKHopSubgraphForRootNodes(chunk, fanouts)):
                out[int(nid)] = rnn
        return out

    @staticmethod
    def _run_node_classification(cfg, rnns, labels):
        unl = [rnns[k].SerializeToString() for k in sorted(rnns)]
        files = {"unlabeled": _write(cfg.unlabeled_tfrecord_uri_prefix, unl)}
        pm = cfg.preprocessed_metadata.nodes[0]
        lab = []
        for k in sorted(rnns):
            r = rnns[k]
            if not r.neighborhood.edges:  # isolated nodes produce no training samples
                continue
            lbs = [wire.Label(label_type=lk, label=labels[lk][k]) for lk in pm.label_keys if k in labels.get(lk, {})]
            if not lbs:
                continue
            lab.append(wire.SupervisedNodeClassificationSample(root_node=r.root_node, neighborhood=r.neighborhood,
                                                               root_node_labels=lbs).SerializeToString())
        files["labeled"] = _write(cfg.labeled_tfrecord_uri_prefix, lab)
        return files

    @staticmethod
    def _run_nablp(cfg, svc: HipKHopSamplerService, rnns):
        """createNodeAnchorBasedLinkPredictionSubgraph (NodeAnchorBasedLinkPredictionTask.scala:146-312):
        neighborhood = array_distinct(root nbhd ++ union of the positives' nbhds) looked up from the cached
        per-node subgraphs; pos_edges = [root -> pos]; hard_neg_edges = neg_edges = []"""
        num_pos = cfg.num_positive_samples
        ids = sorted(rnns)
        roots = np.asarray(ids, dtype=np.uint32)
        pos, cnt = svc.engine.sample_positives(roots, num_pos, sampling_seed=svc.sampling_seed)
        pos = pos.cpu().numpy().view(np.uint32).reshape(len(ids), num_pos)
        cnt = cnt.cpu().numpy()
        samples = []
        for i, r in enumerate(ids):
            if cnt[i] == 0:
                continue  # anchors need at least one positive (out-edge)
            nodes = {n.node_id: n for n in rnns[r].neighborhood.nodes}
            edges = {(e.src_node_id, e.dst_node_id): e for e in rnns[r].neighborhood.edges}
            pos_edges = []
            for p in pos[i][: cnt[i]].tolist():
                pos_edges.append(wire.Edge(src_node_id=int(r), dst_node_id=int(p), condensed_edge_type=0))
                for nn in rnns[p].neighborhood.nodes:
                    nodes.setdefault(nn.node_id, nn)
                for e in rnns[p].neighborhood.edges:
                    edges.setdefault((e.src_node_id, e.dst_node_id), e)
            samples.append(wire.NodeAnchorBasedLinkPredictionSample(
                root_node=rnns[r].root_node, pos_edges=pos_edges,
                neighborhood=wire.Graph(nodes=list(nodes.values()), edges=list(edges.values()))).SerializeToString())
        files = {"node_anchor_based_link_prediction": _write(cfg.nablp_tfrecord_uri_prefix, samples)}
        rn = [rnns[k].SerializeToString() for k in ids]
        for node_type, prefix in cfg.random_negative_tfrecord_uri_prefixes.items():
            files[f"random_negative/{node_type}"] = _write(prefix, rn)
        return files


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
