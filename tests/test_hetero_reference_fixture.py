"""The typed sampler outputs the reference holds as split-generator input (scala_spark35/common/src/test/assets/
split_generator/hetero_node_anchor_based_link_prediction/sgs_output: users / stories, edge types user->story = 0 and
story->user = 1, supervision edge type story -> user; produced by the reference's own heterogeneous sampler) through the
host codec, the native typed collate and the split generator.  CPU only."""
import os
import shutil

import numpy as np
import yaml

from gigl_amd import wire
from gigl_amd.config import tfrecord_files

BASE = "ref_assets/split_generator/hetero_node_anchor_based_link_prediction/sgs_output/"
MAIN = BASE + "node_anchor_based_link_prediction_samples/samples/"
RN = {"user": BASE + "random_negative_rooted_neighborhood_samples/user/samples/",
      "story": BASE + "random_negative_rooted_neighborhood_samples/story/samples/"}
ENDS = [(0, 1), (1, 0)]  # condensed edge type -> (condensed src node type, condensed dst node type)


def _records(golden_dir, rel):
    return [r for f in tfrecord_files(os.path.join(golden_dir, rel)) for r in wire.read_tfrecords(f)]


def test_reference_typed_records_reserialize_byte_for_byte(golden_dir):
    main = _records(golden_dir, MAIN)
    assert len(main) == 18
    for r in main:
        m = wire.NodeAnchorBasedLinkPredictionSample.FromString(r)
        assert m.SerializeToString() == r and m.root_node.condensed_node_type == 1 and m.pos_edges
        nodes = {(x.node_id, x.condensed_node_type) for x in m.neighborhood.nodes}
        assert (m.root_node.node_id, 1) in nodes and len(nodes) == len(m.neighborhood.nodes)
        for e in list(m.pos_edges) + list(m.hard_neg_edges):  # story -> user label edges, targets inside the graph
            assert e.src_node_id == m.root_node.node_id and e.condensed_edge_type == 1 and (e.dst_node_id, 0) in nodes
        for e in m.neighborhood.edges:
            s_t, d_t = ENDS[e.condensed_edge_type]
            assert (e.src_node_id, s_t) in nodes and (e.dst_node_id, d_t) in nodes
    for t, (name, rel) in enumerate(RN.items()):
        recs = _records(golden_dir, rel)
        assert len(recs) == (15, 19)[t]
        for r in recs:
            m = wire.RootedNodeNeighborhood.FromString(r)
            assert m.SerializeToString() == r and m.root_node.condensed_node_type == t


def test_native_typed_collate_of_the_reference_training_samples(golden_dir):
    """gigl_collate_typed_records over the reference's typed NodeAnchorBasedLinkPredictionSample records == the
    reference graph builder's rules applied to the decoded messages: per node type first-seen numbering (root first,
    then the neighbourhood's nodes, sample after sample), per edge type distinct edges sorted by (src, dst), label-edge
    targets resolved to local ids of their node type"""
    from gigl_amd._lib import REC_NODE_ANCHOR_LINK_PRED
    from gigl_amd.batches import collate_serialized_typed
    recs = _records(golden_dir, MAIN)
    out = collate_serialized_typed(recs, REC_NODE_ANCHOR_LINK_PRED, 2, ENDS)
    msgs = [wire.NodeAnchorBasedLinkPredictionSample.FromString(r) for r in recs]
    order = {0: [], 1: []}
    seen = set()
    edges = {0: set(), 1: set()}
    for m in msgs:
        for x in m.neighborhood.nodes:
            key = (x.condensed_node_type or 0, x.node_id)
            if key not in seen:
                seen.add(key)
                order[key[0]].append(x.node_id)
        for e in m.neighborhood.edges:
            edges[e.condensed_edge_type or 0].add((e.src_node_id, e.dst_node_id))
    local = {t: {g: i for i, g in enumerate(ids)} for t, ids in order.items()}
    for t in (0, 1):
        assert sorted(out["node_ids"][t].tolist()) == sorted(order[t])
        got_local = {g: i for i, g in enumerate(out["node_ids"][t].tolist())}
        s_t, d_t = ENDS[t]
        l_src = {g: i for i, g in enumerate(out["node_ids"][s_t].tolist())}
        l_dst = {g: i for i, g in enumerate(out["node_ids"][d_t].tolist())}
        want = sorted((l_src[s], l_dst[d]) for s, d in edges[t])
        assert [tuple(c) for c in out["edge_index"][t].T.tolist()] == want
        assert len(got_local) == len(local[t])
    users = out["node_ids"][0].tolist()
    stories = out["node_ids"][1].tolist()
    for i, m in enumerate(msgs):
        assert out["root_type"][i] == 1 and stories[out["root_local"][i]] == m.root_node.node_id
        pos = out["pos_dst"][out["pos_off"][i]:out["pos_off"][i + 1]]
        assert sorted(users[j] for j in pos) == sorted(e.dst_node_id for e in m.pos_edges)
        neg = out["neg_dst"][out["neg_off"][i]:out["neg_off"][i + 1]]
        assert sorted(users[j] for j in neg) == sorted(e.dst_node_id for e in m.hard_neg_edges)


def test_split_generator_keeps_root_node_types(golden_dir, tmp_path):
    """HeterogeneousNodeAnchorBasedLinkPredictionTaskTest.scala:53-124: after splitting, every RootedNodeNeighborhood
    of a node type's random-negative output still has that node type as its root — here for both node types and all
    three splits, through SplitGenerator.run on the fixture's (re-rooted) config"""
    from gigl_amd.split_generator import TEST, TRAIN, VAL, SplitGenerator
    base = tmp_path / "hsg"
    shutil.copytree(os.path.join(golden_dir, "ref_assets"), base / "ref_assets")
    rn_uri = lambda sp: {t: f"out/{sp}/rn/{t}/neighborhoods-" for t in RN}
    cfg = {
        "graphMetadata": {
            "condensedEdgeTypeMap": {"0": {"srcNodeType": "user", "relation": "to", "dstNodeType": "story"},
                                     "1": {"srcNodeType": "story", "relation": "to", "dstNodeType": "user"}},
            "condensedNodeTypeMap": {"0": "user", "1": "story"},
            "edgeTypes": [{"srcNodeType": "user", "relation": "to", "dstNodeType": "story"},
                          {"srcNodeType": "story", "relation": "to", "dstNodeType": "user"}],
            "nodeTypes": ["user", "story"]},
        "taskMetadata": {"nodeAnchorBasedLinkPredictionTaskMetadata": {"supervisionEdgeTypes": [
            {"srcNodeType": "story", "relation": "to", "dstNodeType": "user"}]}},
        "datasetConfig": {"splitGeneratorConfig": {
            "assignerArgs": {"seed": "42", "test_split": "0.2", "train_split": "0.7", "val_split": "0.1"},
            "assignerClsPath": "splitgenerator.lib.assigners.TransductiveEdgeToLinkSplitHashingAssigner",
            "splitStrategyClsPath": "splitgenerator.lib.split_strategies.TransductiveNodeAnchorBasedLinkPredictionSplitStrategy"}},
        "sharedConfig": {
            "datasetMetadata": {"nodeAnchorBasedLinkPredictionDataset": {
                "trainMainDataUri": "out/train/main/", "valMainDataUri": "out/val/main/", "testMainDataUri": "out/test/main/",
                "trainNodeTypeToRandomNegativeDataUri": rn_uri("train"), "valNodeTypeToRandomNegativeDataUri": rn_uri("val"),
                "testNodeTypeToRandomNegativeDataUri": rn_uri("test")}},
            "flattenedGraphMetadata": {"nodeAnchorBasedLinkPredictionOutput": {
                "tfrecordUriPrefix": MAIN, "nodeTypeToRandomNegativeTfrecordUriPrefix": dict(RN)}}}}
    (base / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    files = SplitGenerator().run("job", "cfg.yaml", None, uri_base=str(base))
    for t, (name, want) in enumerate((("user", 15), ("story", 19))):
        for sp in (TRAIN, VAL, TEST):
            recs = [wire.RootedNodeNeighborhood.FromString(r) for f in files[f"random_negative/{name}"][sp]
                    for r in wire.read_tfrecords(f)]
            assert len(recs) == want and all(m.root_node.condensed_node_type == t for m in recs)
    main = {sp: [wire.NodeAnchorBasedLinkPredictionSample.FromString(r) for f in files["main"][sp]
                 for r in wire.read_tfrecords(f)] for sp in (TRAIN, VAL, TEST)}
    assert len(main[VAL]) == 18 and len(main[TEST]) == 18 and 0 < len(main[TRAIN]) <= 18
    assert all(m.root_node.condensed_node_type == 1 for sp in main for m in main[sp])
    # transductive message passing: a later split's graph contains the earlier ones' edges
    n_edges = {sp: sum(len(m.neighborhood.edges) for m in main[sp]) for sp in main}
    assert n_edges[VAL] <= n_edges[TEST] and np.isfinite(n_edges[TRAIN])
