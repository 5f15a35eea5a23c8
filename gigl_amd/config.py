"""Task / preprocessed-metadata config readers for the hot path.

The reference serialises two protos to YAML — the task config `GbmlConfig`
(proto/snapchat/research/gbml/gbml_config.proto:18-237) and `PreprocessedMetadata`
(proto/snapchat/research/gbml/preprocessed_metadata.proto:5-64) — and wraps them in
`GbmlConfigPbWrapper` (python/gigl/src/common/types/pb_wrappers/gbml_config.py).  This module reads the
SAME YAML documents (camelCase proto-JSON field names) and exposes only the fields this path uses
(SURVEY.md §5.6).  Cloud URIs (gs://, BigQuery) are out of scope: URIs are local paths, optionally
relative to `uri_base`.
"""
from __future__ import annotations

import glob
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import yaml


def _get(d, path, default=None):
    cur = d
    for k in path.split("."):
        if not isinstance(cur, dict) or k not in cur:
            return default
        cur = cur[k]
    return cur


# records per part file the sampler writes (spark-tfrecord part files, TFRecordIO.scala:53-69); the in-HBM route derives
# the TFRecord route's reading order from it (gigl_amd/hbm.py), so both read it HERE at call time
RECORDS_PER_PART_FILE = 100_000


def resolve_uri(uri: str, uri_base: Optional[str]) -> str:
    if uri is None:
        raise ValueError("missing URI")
    if "://" in uri and not uri.startswith("file://"):
        raise NotImplementedError(f"only local URIs are supported on this path (got {uri!r})")
    p = uri[len("file://"):] if uri.startswith("file://") else uri
    if not os.path.isabs(p) and uri_base:
        p = os.path.join(uri_base, p)
    return p


def tfrecord_files(prefix: str) -> List[str]:
    """the reference globs `prefix*.tfrecord`
    (python/gigl/src/common/types/pb_wrappers/dataset_metadata_utils.py:44-63); spark-tfrecord writes
    part files under a directory prefix, so both layouts are accepted"""
    out = sorted(glob.glob(prefix + "*.tfrecord"))
    if not out and os.path.isdir(prefix):
        out = sorted(glob.glob(os.path.join(prefix, "*.tfrecord")))
    return out


@dataclass
class NodeMetadata:
    tfrecord_uri_prefix: str
    node_id_key: str
    feature_keys: List[str]
    label_keys: List[str]
    feature_dim: int


@dataclass
class EdgeInfo:
    """EdgeMetadataInfo (preprocessed_metadata.proto:28-43) of the user-defined positive / negative label edges"""
    tfrecord_uri_prefix: str
    feature_keys: List[str]
    feature_dim: int


@dataclass
class EdgeMetadata:
    tfrecord_uri_prefix: str
    src_node_id_key: str
    dst_node_id_key: str
    feature_keys: List[str]
    feature_dim: int
    positive_edge_info: Optional[EdgeInfo] = None  # :54-57: present = user-defined positives / hard negatives
    negative_edge_info: Optional[EdgeInfo] = None


@dataclass
class PreprocessedMetadata:
    """preprocessed_metadata.proto:5-64 (homogeneous graphs: condensed type '0')"""
    nodes: Dict[int, NodeMetadata] = field(default_factory=dict)
    edges: Dict[int, EdgeMetadata] = field(default_factory=dict)

    @classmethod
    def from_yaml(cls, path: str) -> "PreprocessedMetadata":
        doc = yaml.safe_load(open(path)) or {}
        m = cls()
        for k, v in (doc.get("condensedNodeTypeToPreprocessedMetadata") or {}).items():
            m.nodes[int(k)] = NodeMetadata(
                tfrecord_uri_prefix=v["tfrecordUriPrefix"], node_id_key=v["nodeIdKey"],
                feature_keys=list(v.get("featureKeys") or []), label_keys=list(v.get("labelKeys") or []),
                feature_dim=int(v.get("featureDim", 0)))
        for k, v in (doc.get("condensedEdgeTypeToPreprocessedMetadata") or {}).items():
            main = v.get("mainEdgeInfo") or {}
            def info(d):
                if not d or not d.get("tfrecordUriPrefix"):
                    return None
                return EdgeInfo(tfrecord_uri_prefix=d["tfrecordUriPrefix"], feature_keys=list(d.get("featureKeys") or []),
                                feature_dim=int(d.get("featureDim", 0)))
            m.edges[int(k)] = EdgeMetadata(
                tfrecord_uri_prefix=main["tfrecordUriPrefix"], src_node_id_key=v["srcNodeIdKey"],
                dst_node_id_key=v["dstNodeIdKey"], feature_keys=list(main.get("featureKeys") or []),
                feature_dim=int(main.get("featureDim", 0)), positive_edge_info=info(v.get("positiveEdgeInfo")),
                negative_edge_info=info(v.get("negativeEdgeInfo")))
        return m


class GbmlConfigPbWrapper:
    """the subset of the reference wrapper this path reads; field names per gbml_config.proto"""

    def __init__(self, doc: dict, uri_base: Optional[str] = None):
        self.doc = doc or {}
        self.uri_base = uri_base
        self._preprocessed: Optional[PreprocessedMetadata] = None

    @classmethod
    def from_uri(cls, task_config_uri: str, uri_base: Optional[str] = None) -> "GbmlConfigPbWrapper":
        path = resolve_uri(task_config_uri, uri_base)
        return cls(yaml.safe_load(open(path)), uri_base=uri_base)

    # ---- task
    @property
    def task_kind(self) -> str:
        tm = self.doc.get("taskMetadata") or {}
        if "nodeBasedTaskMetadata" in tm:
            return "node_classification"
        if "nodeAnchorBasedLinkPredictionTaskMetadata" in tm:
            return "node_anchor_based_link_prediction"
        raise ValueError(f"unsupported taskMetadata: {list(tm)}")

    @property
    def is_graph_directed(self) -> bool:
        return bool(_get(self.doc, "sharedConfig.isGraphDirected", False))

    # ---- sampler (gbml_config.proto:72-108; subgraph_sampling_strategy.proto:38-58)
    @property
    def num_hops(self) -> int:
        return int(_get(self.doc, "datasetConfig.subgraphSamplerConfig.numHops", 2) or 2)

    @property
    def num_neighbors_to_sample(self) -> int:
        return int(_get(self.doc, "datasetConfig.subgraphSamplerConfig.numNeighborsToSample", 0) or 0)

    @property
    def fanouts(self) -> List[int]:
        """per-hop fanouts: the SamplingOp chain if a strategy is given (one op per hop, op k's input = op
        k-1), else the legacy single fanout for every hop (SGSPureSparkV1Task.scala:693-705)"""
        strat = _get(self.doc, "datasetConfig.subgraphSamplerConfig.subgraphSamplingStrategy")
        if strat:
            mpps = _get(strat, "messagePassingPaths.paths") or []
            if len(mpps) != 1:
                raise NotImplementedError("homogeneous graphs: exactly one message passing path")
            ops = mpps[0].get("samplingOps") or []
            by_name = {op["opName"]: op for op in ops}
            chain, cur = [], [op for op in ops if not op.get("inputOpNames")]
            if len(cur) != 1:
                raise NotImplementedError("only linear SamplingOp chains are supported")
            op = cur[0]
            while op is not None:
                chain.append(int(_get(op, "randomUniform.numNodesToSample", 0) or op.get("numNodesToSample", 0)))
                nxt = [o for o in ops if (o.get("inputOpNames") or []) == [op["opName"]]]
                if len(nxt) > 1:
                    raise NotImplementedError("only linear SamplingOp chains are supported")
                op = nxt[0] if nxt else None
            assert by_name and all(c > 0 for c in chain)
            return chain
        f = self.num_neighbors_to_sample
        if f <= 0:
            raise ValueError("numNeighborsToSample must be set")
        return [f] * self.num_hops

    @property
    def num_max_training_samples_to_output(self) -> int:
        """0 = no cap (gbml_config.proto:111; downsampleNumberOfNodes, SGSPureSparkV1Task.scala:1042-1081)"""
        return int(_get(self.doc, "datasetConfig.subgraphSamplerConfig.numMaxTrainingSamplesToOutput", 0) or 0)

    @property
    def num_positive_samples(self) -> int:
        return int(_get(self.doc, "datasetConfig.subgraphSamplerConfig.numPositiveSamples", 0) or 0)

    @property
    def num_user_defined_positive_samples(self) -> int:
        return int(_get(self.doc, "datasetConfig.subgraphSamplerConfig.numUserDefinedPositiveSamples", 0) or 0)

    @property
    def num_user_defined_negative_samples(self) -> int:
        return int(_get(self.doc, "datasetConfig.subgraphSamplerConfig.numUserDefinedNegativeSamples", 0) or 0)

    @property
    def experimental_flags(self) -> Dict[str, str]:
        return dict(_get(self.doc, "datasetConfig.subgraphSamplerConfig.experimentalFlags", {}) or {})

    @property
    def permutation_strategy(self) -> str:
        # "deterministic" = the reproducible hash permutation (SamplingStrategy.scala:16-82); other values = F.shuffle.
        # Default as the reference's: experimentalFlags.getOrElse("permutation_strategy", NonDeterministic)
        # (SubgraphSamplerTask.scala:15 / SGSPureSparkV1Task callers)
        return self.experimental_flags.get("permutation_strategy", "non-deterministic")

    # ---- data locations (flattened_graph_metadata.proto:6-39)
    def _fgm(self, path: str) -> Optional[str]:
        v = _get(self.doc, "sharedConfig.flattenedGraphMetadata." + path)
        return resolve_uri(v, self.uri_base) if v else None

    @property
    def labeled_tfrecord_uri_prefix(self):
        return self._fgm("supervisedNodeClassificationOutput.labeledTfrecordUriPrefix")

    @property
    def unlabeled_tfrecord_uri_prefix(self):
        return self._fgm("supervisedNodeClassificationOutput.unlabeledTfrecordUriPrefix")

    @property
    def nablp_tfrecord_uri_prefix(self):
        return self._fgm("nodeAnchorBasedLinkPredictionOutput.tfrecordUriPrefix")

    @property
    def random_negative_tfrecord_uri_prefixes(self) -> Dict[str, str]:
        m = _get(self.doc, "sharedConfig.flattenedGraphMetadata.nodeAnchorBasedLinkPredictionOutput."
                           "nodeTypeToRandomNegativeTfrecordUriPrefix", {}) or {}
        return {k: resolve_uri(v, self.uri_base) for k, v in m.items()}

    # ---- split generator outputs (dataset_metadata.proto: SupervisedNodeClassificationDataset /
    #      NodeAnchorBasedLinkPredictionDataset); None when the config names none
    def dataset_split_uri(self, split: str) -> Optional[str]:
        dm = _get(self.doc, "sharedConfig.datasetMetadata", {}) or {}
        snc = dm.get("supervisedNodeClassificationDataset") or {}
        lp = dm.get("nodeAnchorBasedLinkPredictionDataset") or {}
        v = snc.get(f"{split}DataUri") or lp.get(f"{split}MainDataUri")
        return resolve_uri(v, self.uri_base) if v else None

    def random_negative_split_uris(self, split: str) -> Dict[str, str]:
        lp = _get(self.doc, "sharedConfig.datasetMetadata.nodeAnchorBasedLinkPredictionDataset", {}) or {}
        m = lp.get(f"{split}NodeTypeToRandomNegativeDataUri") or {}
        return {k: resolve_uri(v, self.uri_base) for k, v in m.items()}

    @property
    def preprocessed_metadata(self) -> PreprocessedMetadata:
        if self._preprocessed is None:
            uri = _get(self.doc, "sharedConfig.preprocessedMetadataUri")
            self._preprocessed = PreprocessedMetadata.from_yaml(resolve_uri(uri, self.uri_base))
        return self._preprocessed

    @property
    def node_types(self) -> List[str]:
        return list(_get(self.doc, "graphMetadata.nodeTypes", ["node"]) or ["node"])

    # ---- typed graphs (graph_schema.proto GraphMetadata; gbml_config.proto TaskMetadata)
    @property
    def condensed_node_type_map(self) -> Dict[int, str]:
        m = _get(self.doc, "graphMetadata.condensedNodeTypeMap") or {}
        return {int(k): str(v) for k, v in m.items()} or {0: self.node_types[0]}

    @property
    def condensed_edge_type_map(self) -> Dict[int, tuple]:
        """condensed edge type -> (srcNodeType, relation, dstNodeType)"""
        m = _get(self.doc, "graphMetadata.condensedEdgeTypeMap") or {}
        return {int(k): (str(v["srcNodeType"]), str(v["relation"]), str(v["dstNodeType"])) for k, v in m.items()}

    @property
    def is_heterogeneous(self) -> bool:
        return len(self.condensed_node_type_map) > 1 or len(self.condensed_edge_type_map) > 1

    @property
    def supervision_edge_types(self) -> List[tuple]:
        ets = _get(self.doc, "taskMetadata.nodeAnchorBasedLinkPredictionTaskMetadata.supervisionEdgeTypes") or []
        return [(str(e["srcNodeType"]), str(e["relation"]), str(e["dstNodeType"])) for e in ets]

    @property
    def should_include_isolated_nodes_in_training(self) -> bool:
        return bool(_get(self.doc, "sharedConfig.shouldIncludeIsolatedNodesInTraining", False))

    @property
    def message_passing_paths(self) -> List[dict]:
        """subgraphSamplingStrategy.messagePassingPaths.paths (subgraph_sampling_strategy.proto:38-58): one
        {rootNodeType, samplingOps} entry per root node type; empty when the config carries no strategy"""
        strat = _get(self.doc, "datasetConfig.subgraphSamplerConfig.subgraphSamplingStrategy") or {}
        return list(_get(strat, "messagePassingPaths.paths") or [])

    # ---- plugins (gbml_config.proto:172-237)
    @property
    def trainer_cls_path(self) -> Optional[str]:
        return _get(self.doc, "trainerConfig.trainerClsPath")

    @property
    def trainer_args(self) -> Dict[str, str]:
        return {k: str(v) for k, v in (_get(self.doc, "trainerConfig.trainerArgs", {}) or {}).items()}

    @property
    def inferencer_cls_path(self) -> Optional[str]:
        return _get(self.doc, "inferencerConfig.inferencerClsPath")

    @property
    def inferencer_args(self) -> Dict[str, str]:
        return {k: str(v) for k, v in (_get(self.doc, "inferencerConfig.inferencerArgs", {}) or {}).items()}

    @property
    def inference_batch_size(self) -> int:
        return int(_get(self.doc, "inferencerConfig.inferenceBatchSize", 3000) or 3000)  # v1/lib/utils.py:112

    @property
    def trained_model_uri(self) -> Optional[str]:
        v = _get(self.doc, "sharedConfig.trainedModelMetadata.trainedModelUri")
        return resolve_uri(v, self.uri_base) if v else None

    @property
    def eval_metrics_uri(self) -> Optional[str]:
        v = _get(self.doc, "sharedConfig.trainedModelMetadata.evalMetricsUri")
        return resolve_uri(v, self.uri_base) if v else None

    @property
    def embeddings_output_path(self) -> Optional[str]:
        m = _get(self.doc, "sharedConfig.inferenceMetadata.nodeTypeToInferencerOutputInfoMap", {}) or {}
        for _, v in m.items():
            p = v.get("embeddingsPath") or v.get("predictionsPath")
            if p:
                return resolve_uri(p, self.uri_base)
        return None

    @property
    def should_skip_training(self) -> bool:
        return bool(_get(self.doc, "sharedConfig.shouldSkipTraining", False))

    @property
    def should_skip_model_evaluation(self) -> bool:
        return bool(_get(self.doc, "sharedConfig.shouldSkipModelEvaluation", False))
