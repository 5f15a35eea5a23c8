#!/usr/bin/env python3
"""Generate tests/golden/ from the reference checkout (run in the BUILD container only).

What is produced (all DATA — inputs and expected outputs, never reference source text):
  ref_assets/…/*.tfrecord   byte copies of the data files the reference's own tests hold
                            (scala/common/src/test/assets/{subgraph_sampler,split_generator}/…)
  ref_assets_decoded.json   the same records decoded with the REFERENCE's generated *_pb2 classes
                            (expected output for gigl_amd.wire)
  xxh64_int32.json          XXH64(le32(x), seed) known answers from the canonical xxhash library
  graph_builder_traces.json global->local remap + ordered edges produced by the REFERENCE
                            GraphBuilder (abstract_graph_builder.py, imported) on seeded samples
  eval_metrics.json         hit_rate_at_k / mean_reciprocal_rank outputs of the REFERENCE functions
  toy_graph.json            the 27-node toy graph of gigl/src/mocking/mocking_assets (data only)
"""
import json
import os
import shutil
import struct
import sys

import numpy as np

REF = os.environ.get("GIGL_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, os.path.join(REF, "python"))

ASSETS = [
    "subgraph_sampler/supervised_node_classification/node_data/data.tfrecord",
    "subgraph_sampler/supervised_node_classification/edge_data/data.tfrecord",
    "subgraph_sampler/node_anchor_based_link_prediction/node_data/data.tfrecord",
    "subgraph_sampler/node_anchor_based_link_prediction/edge_data/data.tfrecord",
    "subgraph_sampler/node_anchor_based_link_prediction/user_defined_pos/data.tfrecord",
    "subgraph_sampler/node_anchor_based_link_prediction/user_defined_neg/data.tfrecord",
    # the heterogeneous fixture of the Spark-3.5 graph-DB sampler (two node types, two edge types, tf.Example tables)
    "subgraph_sampler/heterogeneous/node_anchor_based_link_prediction/node_features_dir/user/features/data.tfrecord",
    "subgraph_sampler/heterogeneous/node_anchor_based_link_prediction/node_features_dir/story/features/data.tfrecord",
    "subgraph_sampler/heterogeneous/node_anchor_based_link_prediction/edge_features_dir/user-to-story/main_edges/features/data.tfrecord",
    "subgraph_sampler/heterogeneous/node_anchor_based_link_prediction/edge_features_dir/story-to-user/main_edges/features/data.tfrecord",
    "split_generator/supervised_node_classification/sgs_output/unlabeled/samples/data.tfrecord",
    "split_generator/supervised_node_classification/sgs_output/labeled/samples/data.tfrecord",
    "split_generator/node_anchor_based_link_prediction/sgs_output/random_negative_rooted_neighborhood_samples/user/data.tfrecord",
    "split_generator/node_anchor_based_link_prediction/sgs_output/node_anchor_based_link_prediction_samples/data.tfrecord",
]
# typed sampler outputs the reference holds as split-generator input (scala_spark35 assets; HeterogeneousNodeAnchorBased
# LinkPredictionTaskTest.scala): the part files keep their directory, renamed data.tfrecord
HETERO_SPLIT_ASSETS = [
    "split_generator/hetero_node_anchor_based_link_prediction/sgs_output/node_anchor_based_link_prediction_samples/samples",
    "split_generator/hetero_node_anchor_based_link_prediction/sgs_output/random_negative_rooted_neighborhood_samples/user/samples",
    "split_generator/hetero_node_anchor_based_link_prediction/sgs_output/random_negative_rooted_neighborhood_samples/story/samples",
]


def raw_records(path):
    data = open(path, "rb").read()
    pos = 0
    while pos < len(data):
        (ln,) = struct.unpack("<Q", data[pos:pos + 8])
        pos += 12
        yield data[pos:pos + ln]
        pos += ln + 4


def node_d(n):
    d = {"node_id": n.node_id, "feature_values": [float(np.float32(x)) for x in n.feature_values]}
    if n.HasField("condensed_node_type"):
        d["condensed_node_type"] = n.condensed_node_type
    return d


def edge_d(e):
    d = {"src_node_id": e.src_node_id, "dst_node_id": e.dst_node_id,
         "feature_values": [float(np.float32(x)) for x in e.feature_values]}
    if e.HasField("condensed_edge_type"):
        d["condensed_edge_type"] = e.condensed_edge_type
    return d


def graph_d(g):
    return {"nodes": [node_d(n) for n in g.nodes], "edges": [edge_d(e) for e in g.edges]}


def hetero_sampler_configs(base):
    """the reference's heterogeneous sampler fixture config + its preprocessed metadata as test data, with the asset
    paths re-rooted to tests/golden/ and the outputs to out/ (tests/golden/configs/hetero_nablp_*.yaml)"""
    rel = "subgraph_sampler/heterogeneous/node_anchor_based_link_prediction/"
    pre = "common/src/test/assets/" + rel
    cfg = open(os.path.join(base, rel, "frozen_gbml_config_graphdb_dblp_local.yaml")).read()
    meta = open(os.path.join(base, rel, "preprocessed_metadata.yaml")).read()
    cfg = cfg.replace(pre + "preprocessed_metadata.yaml", "configs/hetero_nablp_preprocessed_metadata.yaml")
    cfg = cfg.replace(pre + "output/", "out/hetero_nablp/")
    meta = meta.replace(pre, "ref_assets/" + rel)
    os.makedirs(os.path.join(OUT, "configs"), exist_ok=True)
    open(os.path.join(OUT, "configs", "hetero_nablp_frozen_gbml_config.yaml"), "w").write(cfg)
    open(os.path.join(OUT, "configs", "hetero_nablp_preprocessed_metadata.yaml"), "w").write(meta)


def hetero_split_assets():
    import glob
    base35 = os.path.join(REF, "scala_spark35/common/src/test/assets")
    for rel in HETERO_SPLIT_ASSETS:
        (src,) = glob.glob(os.path.join(base35, rel, "*.tfrecord"))
        dst = os.path.join(OUT, "ref_assets", rel, "data.tfrecord")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)


def main():
    os.makedirs(OUT, exist_ok=True)
    base = os.path.join(REF, "scala/common/src/test/assets")
    hetero_sampler_configs(base)
    hetero_split_assets()
    for rel in ASSETS:
        dst = os.path.join(OUT, "ref_assets", rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(base, rel), dst)

    # ---- decode with the reference's own generated classes
    from snapchat.research.gbml import training_samples_schema_pb2 as ts

    decoded = {}
    for rel in ASSETS:
        recs = list(raw_records(os.path.join(base, rel)))
        if "unlabeled/samples" in rel or "random_negative" in rel:
            out = []
            for r in recs:
                m = ts.RootedNodeNeighborhood()
                m.ParseFromString(r)
                out.append({"root_node": node_d(m.root_node), "neighborhood": graph_d(m.neighborhood),
                            "reserialized_hex": m.SerializeToString(deterministic=True).hex()})
            decoded[rel] = {"type": "RootedNodeNeighborhood", "records": out}
        elif "labeled/samples" in rel:
            out = []
            for r in recs:
                m = ts.SupervisedNodeClassificationSample()
                m.ParseFromString(r)
                out.append({"root_node": node_d(m.root_node), "neighborhood": graph_d(m.neighborhood),
                            "root_node_labels": [{"label_type": l.label_type, "label": l.label}
                                                 for l in m.root_node_labels],
                            "reserialized_hex": m.SerializeToString(deterministic=True).hex()})
            decoded[rel] = {"type": "SupervisedNodeClassificationSample", "records": out}
        elif "node_anchor_based_link_prediction_samples" in rel:
            out = []
            for r in recs:
                m = ts.NodeAnchorBasedLinkPredictionSample()
                m.ParseFromString(r)
                out.append({"root_node": node_d(m.root_node), "neighborhood": graph_d(m.neighborhood),
                            "pos_edges": [edge_d(e) for e in m.pos_edges],
                            "hard_neg_edges": [edge_d(e) for e in m.hard_neg_edges],
                            "neg_edges": [edge_d(e) for e in m.neg_edges],
                            "reserialized_hex": m.SerializeToString(deterministic=True).hex()})
            decoded[rel] = {"type": "NodeAnchorBasedLinkPredictionSample", "records": out}
        else:
            decoded[rel] = {"type": "tf.Example", "n_records": len(recs)}
    json.dump(decoded, open(os.path.join(OUT, "ref_assets_decoded.json"), "w"), indent=0)

    # ---- XXH64 known answers (canonical library)
    import xxhash

    rng = np.random.default_rng(12345)
    xs = [0, 1, -1, 2, 42, 43, 84, 127, 128, 255, 256, 65535, 65536, 2**31 - 1, -(2**31), 123456789]
    xs += [int(v) for v in rng.integers(-(2**31), 2**31, size=240)]
    vec = [{"x": x, "seed": s, "h": format(xxhash.xxh64(struct.pack("<i", x), seed=s).intdigest(), "016x")}
           for s in (0, 42) for x in xs]
    vec.append({"x": None, "seed": 0, "h": format(xxhash.xxh64(b"", seed=0).intdigest(), "016x")})
    json.dump({"library": "python-xxhash " + xxhash.VERSION + " / xxHash " + xxhash.XXHASH_VERSION,
               "vectors": vec}, open(os.path.join(OUT, "xxh64_int32.json"), "w"), indent=0)

    # ---- GraphBuilder traces from the reference's abstract builder
    from gigl.src.common.graph_builder.abstract_graph_builder import GraphBuilder
    from gigl.src.common.types.graph_data import Edge, EdgeType, Node, NodeId, NodeType, Relation

    class TraceBuilder(GraphBuilder):
        def __init__(self):
            self.reset()

        def build(self):
            return None

    nt = NodeType("n")
    et = EdgeType(nt, Relation("r"), nt)
    traces = []
    for case in range(12):
        r = np.random.default_rng(1000 + case)
        n_samples = int(r.integers(1, 6))
        universe = int(r.integers(4, 40))
        samples = []
        bld = TraceBuilder()
        for _ in range(n_samples):
            k = int(r.integers(1, min(universe, 12) + 1))
            nodes = [int(x) for x in r.choice(universe, size=k, replace=False)]
            ne = int(r.integers(0, 3 * k + 1))
            edges = [[int(nodes[int(r.integers(0, k))]), int(nodes[int(r.integers(0, k))])] for _ in range(ne)]
            samples.append({"nodes": nodes, "edges": edges})
            # == GraphBuilder.add_graph_data order: nodes first, then edges with skip_if_exists
            for v in nodes:
                g = Node(type=nt, id=NodeId(v))
                if g not in bld.global_node_to_subgraph_node_map:
                    bld.add_node(node=g)
            seen_in_sample = set()
            for s, d in edges:
                if (s, d) in seen_in_sample:  # a per-sample PygGraphData holds each edge once
                    continue
                seen_in_sample.add((s, d))
                bld.add_edge(edge=Edge.from_nodes(Node(type=nt, id=NodeId(s)), Node(type=nt, id=NodeId(d)),
                                                  Relation("r")), skip_if_exists=True)
        mapping = {int(g.id): int(l.id) for g, l in bld.global_node_to_subgraph_node_map.items()}
        ordered = [[int(e.src_node_id), int(e.dst_node_id)] for e in bld.ordered_edges[et]]
        traces.append({"samples": samples, "global_to_local": mapping, "ordered_edges_local": ordered})
    json.dump(traces, open(os.path.join(OUT, "graph_builder_traces.json"), "w"))

    # ---- eval metrics from the reference functions
    import torch

    from gigl.src.common.utils.eval_metrics import hit_rate_at_k, mean_reciprocal_rank

    cases = []
    g = torch.Generator().manual_seed(7)
    for npos, nneg in [(1, 1), (1, 5), (3, 10), (2, 600), (5, 3)]:
        pos = torch.rand(npos, generator=g)
        neg = torch.rand(nneg, generator=g)
        ks = torch.tensor([1, 5, 10, 50, 100, 500])
        cases.append({"pos": pos.tolist(), "neg": neg.tolist(), "ks": ks.tolist(),
                      "hits": hit_rate_at_k(pos, neg, ks).tolist(),
                      "mrr": float(mean_reciprocal_rank(pos, neg))})
    json.dump(cases, open(os.path.join(OUT, "eval_metrics.json"), "w"))

    # ---- toy graph (data)
    import yaml

    toy = yaml.safe_load(open(os.path.join(REF, "python/gigl/src/mocking/mocking_assets/toy_graph_data.yaml")))
    edges = [[row["src"], d] for row in toy["adj_list"]["user_friend_user"] for d in row["dst"]]
    nodes = [{"id": row["src"], "features": row["features"]} for row in toy["nodes"]["user"]]
    json.dump({"edges": edges, "nodes": nodes}, open(os.path.join(OUT, "toy_graph.json"), "w"))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
