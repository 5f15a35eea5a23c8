"""split generator: MurmurHash3 restatement vs the public MurmurHash3_x86_32 vectors, the reference's property tests
restated (scala/split_generator/src/test/scala/TransductiveEdgeToLinkSplitHashingAssignerTest.scala,
SupervisedNodeClassificationSplitStrategyTest.scala, TransductiveNodeAnchorBasedLinkPredictionSplitStrategyTest.scala)
and an end-to-end run over the reference's real sampler outputs.  Host code only."""
import os
import shutil

import numpy as np
import yaml

from gigl_amd import wire
from gigl_amd.config import tfrecord_files
from gigl_amd.split_generator import (MESSAGE, MESSAGE_AND_SUPERVISION, SUPERVISION, TEST, TRAIN, VAL,
                                      InductiveSupervisedNodeClassificationSplitStrategy,
                                      NodeToDatasetSplitHashingAssigner, SplitGenerator,
                                      TransductiveEdgeToLinkSplitHashingAssigner,
                                      TransductiveNodeAnchorBasedLinkPredictionSplitStrategy,
                                      TransductiveSupervisedNodeClassificationSplitStrategy, murmur3_bytes_hash)

SG = "ref_assets/split_generator"


def test_murmur3_x86_32_known_answers():
    u = lambda b, seed: murmur3_bytes_hash(b, seed) & 0xFFFFFFFF
    assert u(b"", 0) == 0
    assert u(b"", 1) == 0x514E28B7
    assert u(b"", 0xFFFFFFFF) == 0x81F16F39
    assert u(b"\xff\xff\xff\xff", 0) == 0x76293B50
    assert u(b"\x21\x43\x65\x87", 0) == 0xF55B516B
    assert u(b"\x21\x43\x65\x87", 0x5082EDEE) == 0x2362F9DE
    assert u(b"\x21\x43\x65", 0) == 0x7E4A8634
    assert u(b"\x21\x43", 0) == 0xA0F7B07A
    assert u(b"\x21", 0) == 0x72661CF4
    assert u(b"\x00\x00\x00\x00", 0) == 0x2362F9DE
    assert u(b"test", 0) == 0xBA6BD213
    assert u(b"Hello, world!", 0) == 0xC0363E43
    assert u(b"The quick brown fox jumps over the lazy dog", 0) == 0x2E4FF723
    assert -2**31 <= murmur3_bytes_hash(b"12-0-7") < 2**31  # signed like Scala's Int
    # the oracle's C restatement is pinned on the same public vectors and agrees with the host routine on keys
    import oracle.oracle as orc
    for data, seed, want in ((b"", 1, 0x514E28B7), (b"\x21\x43\x65\x87", 0x5082EDEE, 0x2362F9DE), (b"test", 0, 0xBA6BD213),
                             (b"Hello, world!", 0, 0xC0363E43),
                             (b"The quick brown fox jumps over the lazy dog", 0, 0x2E4FF723), (b"\x21\x43\x65", 0, 0x7E4A8634)):
        assert orc.murmur3_x86_32(data, seed) & 0xFFFFFFFF == want
    ids = np.array([0, 7, 12, 4096, 123456789, 4294967295], dtype=np.uint32)
    from gigl_amd.split_generator import HASH_SPACE_GRANULARITY, SCALA_ARRAY_SEED, edge_unique_id, node_unique_id
    assert orc.split_slots(ids, condensed_type=1).tolist() == [
        murmur3_bytes_hash(node_unique_id(int(x), 1), SCALA_ARRAY_SEED) % HASH_SPACE_GRANULARITY for x in ids]
    assert orc.split_slots(ids, ids[::-1], condensed_type=0, symmetric=True).tolist() == [
        murmur3_bytes_hash(edge_unique_id(min(int(x), int(y)), max(int(x), int(y)), 0), SCALA_ARRAY_SEED)
        % HASH_SPACE_GRANULARITY for x, y in zip(ids, ids[::-1])]


def _mock_edges(n=100):
    """SplitGeneratorTestUtils.getMockEdgePbWrappers: pairs a->b, b->a"""
    rng = np.random.default_rng(0)
    out = []
    for _ in range(n // 2):
        a, b = (int(x) for x in rng.choice(10_000, 2, replace=False))
        out += [wire.Edge(src_node_id=a, dst_node_id=b, condensed_edge_type=0),
                wire.Edge(src_node_id=b, dst_node_id=a, condensed_edge_type=0)]
    return out


ARGS = {"train_split": "0.5", "val_split": "0.25", "test_split": "0.25"}


def test_edge_assigner_undirected_non_disjoint():
    a = TransductiveEdgeToLinkSplitHashingAssigner(ARGS)
    assert a.indices == [0, 5000, 7500, 10000]
    edges = _mock_edges()
    got = [a.assign(e) for e in edges]
    assert set(got) == {(TRAIN, MESSAGE_AND_SUPERVISION), (VAL, MESSAGE_AND_SUPERVISION), (TEST, MESSAGE_AND_SUPERVISION)}
    assert got == [a.assign(e) for e in edges]  # deterministic
    n = {s: sum(1 for g in got if g[0] == s) for s in (TRAIN, VAL, TEST)}
    assert n[TRAIN] > n[VAL] and n[TRAIN] > n[TEST]
    assert all(got[i] == got[i + 1] for i in range(0, len(got), 2))  # a->b and b->a land together


def test_edge_assigner_directed_and_disjoint():
    edges = _mock_edges()
    d = TransductiveEdgeToLinkSplitHashingAssigner({**ARGS, "should_split_edges_symmetrically": "False"})
    got = [d.assign(e) for e in edges]
    assert any(got[i] != got[i + 1] for i in range(0, len(got), 2))
    j = TransductiveEdgeToLinkSplitHashingAssigner({**ARGS, "disjoint_train_ratio": "0.5"})
    assert j.indices == [0, 2500, 5000, 7500, 10000]
    assert {j.assign(e) for e in _mock_edges(400)} == {(TRAIN, MESSAGE), (TRAIN, SUPERVISION),
                                                        (VAL, MESSAGE_AND_SUPERVISION), (TEST, MESSAGE_AND_SUPERVISION)}


def test_default_weights_round_like_scala_float32():
    a = NodeToDatasetSplitHashingAssigner({})
    assert a.indices == [0, 8000, 9000, 10000]
    b = TransductiveEdgeToLinkSplitHashingAssigner({"train_split": "0.7", "val_split": "0.1", "test_split": "0.2"})
    assert b.indices == [0, 7000, 8000, 10000]


def _snc_samples(golden_dir):
    f = os.path.join(golden_dir, SG, "supervised_node_classification/sgs_output/labeled/samples/data.tfrecord")
    return [wire.SupervisedNodeClassificationSample.FromString(r) for r in wire.read_tfrecords(f)]


def test_node_classification_strategies(golden_dir):
    samples = _snc_samples(golden_dir)
    assigner = NodeToDatasetSplitHashingAssigner({"train_split": "0.4", "val_split": "0.3", "test_split": "0.3"})
    for cls in (TransductiveSupervisedNodeClassificationSplitStrategy, InductiveSupervisedNodeClassificationSplitStrategy):
        strat = cls({}, assigner)
        for s in samples:
            outs = {sp: strat.split_training_sample(s, sp) for sp in (TRAIN, VAL, TEST)}
            assert sum(len(v) for v in outs.values()) == 1  # a sample goes to exactly one of train/val/test
            sp = next(k for k, v in outs.items() if v)
            o = outs[sp][0]
            assert assigner.assign(o.root_node) == sp
            if cls is TransductiveSupervisedNodeClassificationSplitStrategy:
                assert o.SerializeToString() == s.SerializeToString()  # the whole graph stays visible
            else:  # inductive: only same-split nodes and edges between them survive
                ids = {n.node_id for n in o.neighborhood.nodes}
                assert all(assigner.assign(n) == sp for n in o.neighborhood.nodes)
                assert all(e.src_node_id in ids and e.dst_node_id in ids for e in o.neighborhood.edges)
                assert o.root_node_labels == s.root_node_labels or [l.label for l in o.root_node_labels] == [l.label for l in s.root_node_labels]


def _nablp_samples(golden_dir):
    f = os.path.join(golden_dir, SG, "node_anchor_based_link_prediction/sgs_output/node_anchor_based_link_prediction_samples/data.tfrecord")
    return [wire.NodeAnchorBasedLinkPredictionSample.FromString(r) for r in wire.read_tfrecords(f)]


def test_link_prediction_strategy_rules(golden_dir):
    samples = _nablp_samples(golden_dir)
    for disjoint in (False, True):
        args = {"train_split": "0.5", "val_split": "0.25", "test_split": "0.25"}
        if disjoint:
            args["disjoint_train_ratio"] = "0.5"
        assigner = TransductiveEdgeToLinkSplitHashingAssigner(args)
        strat = TransductiveNodeAnchorBasedLinkPredictionSplitStrategy({"is_disjoint_mode": str(disjoint).lower()}, assigner)
        seen_pos = 0
        for s in samples:
            placed = 0
            for sp in (TRAIN, VAL, TEST):
                outs = strat.split_training_sample(s, sp)
                assert len(outs) <= 1
                if not outs:
                    assert sp == TRAIN  # only a train sample without positives is dropped
                    continue
                o = outs[0]
                placed += len(o.pos_edges)
                if sp == TRAIN:
                    assert o.pos_edges
                for e in o.pos_edges + o.hard_neg_edges + o.neg_edges:
                    ds, usage = assigner.assign(e)
                    assert ds == sp and not (sp == TRAIN and usage == MESSAGE)
                for e in o.neighborhood.edges:  # message passing visibility
                    ds, usage = assigner.assign(e)
                    if sp == TRAIN:
                        assert ds == TRAIN and (not disjoint or usage == MESSAGE)
                    elif sp == VAL:
                        assert ds == TRAIN
                    else:
                        assert ds in (TRAIN, VAL)
                ids = {n.node_id for n in o.neighborhood.nodes}
                assert o.root_node.node_id in ids
                for e in o.neighborhood.edges + o.pos_edges:
                    assert e.src_node_id in ids and e.dst_node_id in ids
                feats = {n.node_id: n for n in s.neighborhood.nodes}
                assert all(n == feats[n.node_id] for n in o.neighborhood.nodes)  # features carried over
            seen_pos += placed
            n_msg_only = sum(1 for e in s.pos_edges if assigner.assign(e) == (TRAIN, MESSAGE))
            assert placed == len(s.pos_edges) - n_msg_only  # every supervision edge lands in exactly one split
        assert seen_pos > 0


def test_user_defined_labels_strategy_and_assigner(golden_dir):
    """UserDefinedLabelsNodeAnchorBasedLinkPredictionSplitStrategyTest.scala: the message-passing edges are the same in
    every split (and equal the input's), rooted neighbourhoods pass through whole, the assigner only ever answers
    SUPERVISION buckets; plus: every label edge lands in exactly one split and a train sample keeps a positive"""
    from gigl_amd.split_generator import (UserDefinedLabelsEdgeToLinkSplitHashingAssigner,
                                          UserDefinedLabelsNodeAnchorBasedLinkPredictionSplitStrategy, build_strategy)
    a = UserDefinedLabelsEdgeToLinkSplitHashingAssigner(ARGS)
    assert a.indices == [0, 5000, 7500, 10000] and not a.symmetric
    got = {a.assign(e) for e in _mock_edges()}
    assert got == {(TRAIN, SUPERVISION), (VAL, SUPERVISION), (TEST, SUPERVISION)}
    sym = UserDefinedLabelsEdgeToLinkSplitHashingAssigner({**ARGS, "should_split_edges_symmetrically": "True"})
    edges = _mock_edges()
    assert all(sym.assign(edges[i]) == sym.assign(edges[i + 1]) for i in range(0, len(edges), 2))
    assert any(a.assign(edges[i]) != a.assign(edges[i + 1]) for i in range(0, len(edges), 2))
    strat = UserDefinedLabelsNodeAnchorBasedLinkPredictionSplitStrategy(
        {}, UserDefinedLabelsEdgeToLinkSplitHashingAssigner({"train_split": "0.8", "val_split": "0.1", "test_split": "0.1"}))
    samples = _nablp_samples(golden_dir)
    # give the fixture samples hard negatives too (root -> some neighbourhood node)
    for s in samples:
        s.hard_neg_edges = [wire.Edge(src_node_id=s.root_node.node_id, dst_node_id=n.node_id, condensed_edge_type=0)
                            for n in s.neighborhood.nodes[:2]]
    n_train = 0
    for s in samples:
        placed = {"pos": 0, "hard": 0}
        for sp in (TRAIN, VAL, TEST):
            outs = strat.split_training_sample(s, sp)
            if not outs:
                assert sp == TRAIN and not [e for e in s.pos_edges if strat.assigner.assign(e)[0] == TRAIN]
                continue
            o = outs[0]
            assert o.neighborhood == s.neighborhood and o.root_node == s.root_node  # no masking of message edges
            assert all(strat.assigner.assign(e) == (sp, SUPERVISION) for e in o.pos_edges + o.hard_neg_edges)
            placed["pos"] += len(o.pos_edges)
            placed["hard"] += len(o.hard_neg_edges)
            n_train += sp == TRAIN
        kept_pos = len(s.pos_edges) if any(strat.assigner.assign(e)[0] == TRAIN for e in s.pos_edges) else \
            sum(1 for e in s.pos_edges if strat.assigner.assign(e)[0] != TRAIN)
        assert placed["pos"] == kept_pos
    assert n_train > 0
    f = os.path.join(golden_dir, SG, "node_anchor_based_link_prediction/sgs_output/"
                                     "random_negative_rooted_neighborhood_samples/user/data.tfrecord")
    for r in wire.read_tfrecords(f):
        m = wire.RootedNodeNeighborhood.FromString(r)
        for sp in (TRAIN, VAL, TEST):
            assert strat.split_rooted_node_neighborhood_training_sample(m, sp) == [m]
    from gigl_amd.config import GbmlConfigPbWrapper
    cfg = GbmlConfigPbWrapper({"datasetConfig": {"splitGeneratorConfig": {
        "assignerClsPath": "splitgenerator.lib.assigners.UserDefinedLabelsEdgeToLinkSplitHashingAssigner",
        "splitStrategyClsPath": "splitgenerator.lib.split_strategies.UserDefinedLabelsNodeAnchorBasedLinkPredictionSplitStrategy",
        "assignerArgs": {"train_split": 0.6}}}})
    assert isinstance(build_strategy(cfg), UserDefinedLabelsNodeAnchorBasedLinkPredictionSplitStrategy)


def test_udl_anchor_based_supervision_edge_strategy_known_answers():
    """UDLAnchorBasedSupervisionEdgeSplitStrategyTest.scala:71-151, restated case by case"""
    from gigl_amd.split_generator import NodeToDatasetSplitHashingAssigner, UDLAnchorBasedSupervisionEdgeSplitStrategy
    one = np.array([1.0], np.float32)

    def rnn(root, edge_list):
        ids = []
        for s_, d_ in edge_list:
            for v in (s_, d_):
                if v not in ids:
                    ids.append(v)
        return wire.RootedNodeNeighborhood(
            root_node=wire.Node(node_id=root, feature_values=one),
            neighborhood=wire.Graph(nodes=[wire.Node(node_id=v, feature_values=one) for v in ids],
                                    edges=[wire.Edge(src_node_id=a, dst_node_id=b, feature_values=one) for a, b in edge_list]))

    def nalp(root, edge_list, pos, neg):
        r = rnn(root, edge_list)
        mk = lambda lst: [wire.Edge(src_node_id=a, dst_node_id=b, feature_values=one) for a, b in lst]
        return wire.NodeAnchorBasedLinkPredictionSample(root_node=r.root_node, neighborhood=r.neighborhood,
                                                        pos_edges=mk(pos), neg_edges=mk(neg))
    strat = UDLAnchorBasedSupervisionEdgeSplitStrategy(
        {}, NodeToDatasetSplitHashingAssigner({"train_split": "1.0", "val_split": "0.0", "test_split": "0.0"}))
    # every neighbourhood edge is a label edge or the reverse of one: nothing left for message passing -> dropped
    assert strat.split_training_sample(nalp(1, [(1, 2), (2, 1), (3, 1)], [(1, 2)], [(1, 3)]), TRAIN) == []
    # no positive to begin with
    assert strat.split_training_sample(nalp(1, [(1, 2), (1, 5)], [], [(1, 3)]), TRAIN) == []
    sample = nalp(1, [(1, 2), (1, 5), (3, 2), (10, 5)], [(1, 2)], [(3, 2)])
    out = strat.split_training_sample(sample, TRAIN)
    assert out == [nalp(1, [(1, 5), (10, 5)], [(1, 2)], [(3, 2)])]
    assert strat.split_training_sample(sample, TEST) == []  # train_split = 1.0: the anchor is a train node
    r = rnn(1, [(1, 2)])
    for sp in (TRAIN, TEST):  # rooted neighbourhoods are emitted for every split, unchanged
        assert strat.split_rooted_node_neighborhood_training_sample(r, sp) == [r]
    # the filters can be switched off per split, and the reverse direction kept
    keep = UDLAnchorBasedSupervisionEdgeSplitStrategy(
        {"should_filter_train": "false"}, NodeToDatasetSplitHashingAssigner({"train_split": "1.0", "val_split": "0.0",
                                                                            "test_split": "0.0"}))
    assert keep.split_training_sample(sample, TRAIN) == [sample]
    fwd_only = UDLAnchorBasedSupervisionEdgeSplitStrategy(
        {"should_filter_reverse_supervision_edge": "false"},
        NodeToDatasetSplitHashingAssigner({"train_split": "1.0", "val_split": "0.0", "test_split": "0.0"}))
    got = fwd_only.split_training_sample(nalp(1, [(1, 2), (2, 1), (3, 1)], [(1, 2)], [(1, 3)]), TRAIN)
    assert [(e.src_node_id, e.dst_node_id) for e in got[0].neighborhood.edges] == [(2, 1), (3, 1)]
    assert [n.node_id for n in got[0].neighborhood.nodes] == [1, 2, 3]  # the sample's own node order is kept


def test_end_to_end_over_reference_sampler_outputs(golden_dir, tmp_path):
    base = tmp_path / "sg"
    shutil.copytree(os.path.join(golden_dir, "ref_assets"), base / "ref_assets")
    shutil.copytree(os.path.join(golden_dir, "configs"), base / "configs")
    root = "ref_assets/split_generator/node_anchor_based_link_prediction/"
    cfg = {
        "graphMetadata": {"edgeTypes": [{"dstNodeType": "user", "relation": "friend", "srcNodeType": "user"}], "nodeTypes": ["user"]},
        "taskMetadata": {"nodeAnchorBasedLinkPredictionTaskMetadata": {"supervisionEdgeTypes": [
            {"dstNodeType": "user", "relation": "friend", "srcNodeType": "user"}]}},
        "datasetConfig": {"splitGeneratorConfig": {
            "assignerArgs": {"seed": "42", "test_split": "0.2", "train_split": "0.7", "val_split": "0.1"},
            "assignerClsPath": "splitgenerator.lib.assigners.TransductiveEdgeToLinkSplitHashingAssigner",
            "splitStrategyClsPath": "splitgenerator.lib.split_strategies.TransductiveNodeAnchorBasedLinkPredictionSplitStrategy"}},
        "sharedConfig": {
            "datasetMetadata": {"nodeAnchorBasedLinkPredictionDataset": {
                "trainMainDataUri": "out/train/main/", "valMainDataUri": "out/val/main/", "testMainDataUri": "out/test/main/",
                "trainNodeTypeToRandomNegativeDataUri": {"user": "out/train/rn/neighborhoods-"},
                "valNodeTypeToRandomNegativeDataUri": {"user": "out/val/rn/neighborhoods-"},
                "testNodeTypeToRandomNegativeDataUri": {"user": "out/test/rn/neighborhoods-"}}},
            "flattenedGraphMetadata": {"nodeAnchorBasedLinkPredictionOutput": {
                "tfrecordUriPrefix": root + "sgs_output/node_anchor_based_link_prediction_samples/",
                "nodeTypeToRandomNegativeTfrecordUriPrefix": {"user": root + "sgs_output/random_negative_rooted_neighborhood_samples/user/"}}}}}
    (base / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    files = SplitGenerator().run("job", "cfg.yaml", None, uri_base=str(base))
    n_in = len(_nablp_samples(golden_dir))
    main = {sp: [wire.NodeAnchorBasedLinkPredictionSample.FromString(r) for f in files["main"][sp] for r in wire.read_tfrecords(f)]
            for sp in (TRAIN, VAL, TEST)}
    assert len(main[VAL]) == n_in and len(main[TEST]) == n_in and 0 < len(main[TRAIN]) <= n_in
    rn = {sp: [r for f in files["random_negative/user"][sp] for r in wire.read_tfrecords(f)] for sp in (TRAIN, VAL, TEST)}
    assert len(rn[TRAIN]) == len(rn[VAL]) == len(rn[TEST]) == 16  # one per input neighbourhood and split
    assert tfrecord_files(str(base / "out/train/rn/neighborhoods-"))  # prefix-style URIs are honoured
