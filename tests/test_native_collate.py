"""native collate (host C++ in libgigl_hip.so, gigl_collate_records) vs the Python restatement of the reference's
GraphBuilder / collate loops (gigl_amd.batches.build_batch_graph, pinned by tests/golden/graph_builder_traces.json
and the reference's batching tests), on the reference's REAL sampler output files and on synthetic samples.
Host code only: runs without a GPU."""
import os

import numpy as np
import pytest
import torch

from gigl_amd import wire
from gigl_amd.batches import (NodeAnchorBasedLinkPredictionBatch, RootedNodeNeighborhoodBatch,
                              SupervisedNodeClassificationBatch)

SG = "ref_assets/split_generator"


def _records(golden_dir, rel):
    return list(wire.read_tfrecords(os.path.join(golden_dir, SG, rel)))


def _same_graph(a, b):
    assert torch.equal(a.graph.x, b.graph.x)
    assert torch.equal(a.graph.edge_index, b.graph.edge_index)


def test_rooted_neighborhoods_of_the_reference_sampler(golden_dir):
    recs = _records(golden_dir, "supervised_node_classification/sgs_output/unlabeled/samples/data.tfrecord")
    assert len(recs) == 16
    for lo, hi in ((0, 16), (0, 5), (5, 6), (3, 3)):
        nat = RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(recs[lo:hi])
        ref = RootedNodeNeighborhoodBatch.collate_pyg_rooted_node_neighborhood_minibatch(
            [wire.RootedNodeNeighborhood.FromString(r) for r in recs[lo:hi]])
        _same_graph(nat, ref)
        assert torch.equal(nat.condensed_node_type_to_root_node_indices_map[0],
                           ref.condensed_node_type_to_root_node_indices_map[0])
        assert nat.root_nodes == ref.root_nodes
        assert (nat.condensed_node_type_to_subgraph_id_to_global_node_id
                == ref.condensed_node_type_to_subgraph_id_to_global_node_id)


def test_labeled_samples_of_the_reference_sampler(golden_dir):
    recs = _records(golden_dir, "supervised_node_classification/sgs_output/labeled/samples/data.tfrecord")
    nat = SupervisedNodeClassificationBatch.process_raw_pyg_samples_and_collate_fn(recs)
    ref = SupervisedNodeClassificationBatch.collate_pyg_node_classification_minibatch(
        [wire.SupervisedNodeClassificationSample.FromString(r) for r in recs])
    _same_graph(nat, ref)
    assert torch.equal(nat.root_node_indices, ref.root_node_indices)
    assert torch.equal(nat.root_node_labels, ref.root_node_labels)
    assert nat.root_nodes == ref.root_nodes


def test_link_prediction_samples_of_the_reference_sampler(golden_dir):
    d = os.path.join(golden_dir, SG, "node_anchor_based_link_prediction/sgs_output")
    files = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".tfrecord")]
    recs = [rec for f in sorted(files) if "node_anchor" in f.replace(d, "") or "main" in f
            for rec in wire.read_tfrecords(f)]
    if not recs:  # layout fallback: take whichever file decodes as NABLP samples with positives
        for f in sorted(files):
            rs = list(wire.read_tfrecords(f))
            if rs and wire.NodeAnchorBasedLinkPredictionSample.FromString(rs[0]).pos_edges:
                recs = rs
                break
    samples = [wire.NodeAnchorBasedLinkPredictionSample.FromString(r) for r in recs]
    samples = [s for s in samples if s.neighborhood is not None]
    recs = [s.SerializeToString() for s in samples]
    assert recs
    nat = NodeAnchorBasedLinkPredictionBatch.process_raw_pyg_samples_and_collate_fn(recs)
    ref = NodeAnchorBasedLinkPredictionBatch.collate_pyg_node_anchor_based_link_prediction_minibatch(samples)
    _same_graph(nat, ref)
    assert torch.equal(nat.root_node_indices, ref.root_node_indices)
    for which in ("pos_supervision_edge_data", "hard_neg_supervision_edge_data"):
        a, b = getattr(nat, which)[0].root_node_to_target_node_id, getattr(ref, which)[0].root_node_to_target_node_id
        assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


def _rand_sample(rng, n_ids, d, with_label=True, unpacked=False):
    ids = rng.choice(n_ids, size=rng.integers(1, 12), replace=False)
    feat = lambda v: FEATS[v, :d]
    nodes = [wire.Node(node_id=int(v), condensed_node_type=0, feature_values=feat(v)) for v in ids]
    m = rng.integers(0, 20)
    edges = [wire.Edge(src_node_id=int(rng.choice(ids)), dst_node_id=int(rng.choice(ids)), condensed_edge_type=0)
             for _ in range(m)]
    root = nodes[int(rng.integers(0, len(nodes)))]
    labels = [wire.Label(label_type="y", label=int(rng.integers(-3, 9)))] if with_label else []
    return wire.SupervisedNodeClassificationSample(root_node=root, neighborhood=wire.Graph(nodes=nodes, edges=edges),
                                                   root_node_labels=labels)


FEATS = np.random.default_rng(1).standard_normal((500, 9)).astype(np.float32)


@pytest.mark.parametrize("d", [0, 1, 9])
def test_synthetic_batches_incl_duplicates_and_threads(d):
    rng = np.random.default_rng(3)
    samples = [_rand_sample(rng, 500, d) for _ in range(300)]
    samples[17].root_node_labels = [wire.Label(label_type="y", label=0), wire.Label(label_type="z", label=5)]
    recs = [s.SerializeToString() for s in samples]
    nat = SupervisedNodeClassificationBatch.process_raw_pyg_samples_and_collate_fn(recs)
    ref = SupervisedNodeClassificationBatch.collate_pyg_node_classification_minibatch(samples)
    _same_graph(nat, ref)
    assert torch.equal(nat.root_node_indices, ref.root_node_indices)
    assert torch.equal(nat.root_node_labels, ref.root_node_labels)
    # one sample without a label -> labels are dropped for the whole batch, like the reference's all(...)
    samples[5].root_node_labels = []
    recs[5] = samples[5].SerializeToString()
    assert SupervisedNodeClassificationBatch.process_raw_pyg_samples_and_collate_fn(recs).root_node_labels is None


def test_error_cases_match_the_python_loops():
    a = wire.Node(node_id=1, condensed_node_type=0, feature_values=np.array([1.0, 2.0], np.float32))
    a2 = wire.Node(node_id=1, condensed_node_type=0, feature_values=np.array([1.0, 2.5], np.float32))
    a_close = wire.Node(node_id=1, condensed_node_type=0, feature_values=np.array([1.0, 2.0 + 1e-6], np.float32))
    b = wire.Node(node_id=2, condensed_node_type=0, feature_values=np.array([3.0, 4.0], np.float32))
    mk = lambda nodes, edges, root: wire.RootedNodeNeighborhood(
        root_node=root, neighborhood=wire.Graph(nodes=nodes, edges=edges)).SerializeToString()
    fn = RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn
    with pytest.raises(AssertionError):  # abstract_graph_builder.py:66-86
        fn([mk([a, b], [], a), mk([a2], [], a2)])
    fn([mk([a, b], [], a), mk([a_close], [], a_close)])  # allclose -> accepted, first features kept
    with pytest.raises(TypeError):  # :26-30
        fn([mk([a], [wire.Edge(src_node_id=1, dst_node_id=9, condensed_edge_type=0)], a)])
    with pytest.raises(KeyError):
        fn([mk([b], [], a)])
    with pytest.raises(ValueError):
        fn([b"\x0a\xff\xff"])
    # a RootedNodeNeighborhood without a neighborhood collates as the root alone
    lone = wire.RootedNodeNeighborhood(root_node=a).SerializeToString()
    out = fn([lone])
    assert out.graph.x.shape == (1, 2) and out.graph.edge_index.shape == (2, 0)


# ---- edge features (Edge.feature_values -> edge_attr): pinned by traces of the reference GraphBuilder
def _edge_feature_traces(golden_dir):
    import json
    return json.load(open(os.path.join(golden_dir, "graph_builder_edge_feature_traces.json")))


def test_edge_features_match_reference_graph_builder_traces(golden_dir):
    """tests/golden/graph_builder_edge_feature_traces.json (scripts/make_golden_edge_features.py imports the
    reference's abstract_graph_builder.py): the registration that is kept for an edge seen in several samples"""
    doc = _edge_feature_traces(golden_dir)
    for t in doc["traces"]:
        de = t["edge_dim"]
        samples = []
        for smp in t["samples"]:
            nodes = [wire.Node(node_id=v, condensed_node_type=0, feature_values=np.array([v], np.float32))
                     for v in smp["nodes"]]
            edges = [wire.Edge(src_node_id=s, dst_node_id=d, condensed_edge_type=0,
                               feature_values=np.array(f, np.float32)) for s, d, f in smp["edges"]]
            samples.append(wire.RootedNodeNeighborhood(root_node=nodes[0],
                                                       neighborhood=wire.Graph(nodes=nodes, edges=edges)))
        want = {(s, d): np.array(f, np.float32)
                for (s, d), f in zip(map(tuple, t["ordered_edges_local"]), t["ordered_edge_features"])}
        ref = RootedNodeNeighborhoodBatch.collate_pyg_rooted_node_neighborhood_minibatch(samples)
        nat = RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn([s.SerializeToString() for s in samples])
        g2l = {int(k): v for k, v in t["global_to_local"].items()}
        for b in (ref, nat):
            assert {g: l for l, g in b.condensed_node_type_to_subgraph_id_to_global_node_id[0].items()} == g2l
            ei = b.graph.edge_index.numpy()
            assert ei.shape[1] == len(want)
            if not want:
                assert b.graph.edge_attr is None or b.graph.edge_attr.shape[0] == 0
                continue
            assert b.graph.edge_attr.shape == (len(want), de)
            for k in range(ei.shape[1]):
                np.testing.assert_array_equal(b.graph.edge_attr[k].numpy(), want[(int(ei[0, k]), int(ei[1, k]))])
        _same_graph(nat, ref)
        if want:
            assert torch.equal(nat.graph.edge_attr, ref.graph.edge_attr)


def test_mixed_edge_feature_registration_raises_like_the_reference(golden_dir):
    doc = _edge_feature_traces(golden_dir)
    assert all(c["raises"] == "TypeError" for c in doc["mixed_registration"])
    a = wire.Node(node_id=1, condensed_node_type=0, feature_values=np.array([1.0], np.float32))
    b = wire.Node(node_id=2, condensed_node_type=0, feature_values=np.array([2.0], np.float32))
    f = np.array([1.0, 2.0], np.float32)
    for first_has in (True, False):
        edges = [wire.Edge(src_node_id=1, dst_node_id=2, condensed_edge_type=0,
                           feature_values=f if first_has else wire._EMPTY_F32),
                 wire.Edge(src_node_id=2, dst_node_id=1, condensed_edge_type=0,
                           feature_values=wire._EMPTY_F32 if first_has else f)]
        s = wire.RootedNodeNeighborhood(root_node=a, neighborhood=wire.Graph(nodes=[a, b], edges=edges))
        with pytest.raises(TypeError):
            RootedNodeNeighborhoodBatch.collate_pyg_rooted_node_neighborhood_minibatch([s])
        with pytest.raises(TypeError):
            RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn([s.SerializeToString()])


def test_edge_features_random_batches_native_vs_python_loops():
    rng = np.random.default_rng(11)
    samples = []
    for _ in range(200):
        ids = rng.choice(300, size=rng.integers(1, 10), replace=False)
        nodes = [wire.Node(node_id=int(v), condensed_node_type=0, feature_values=FEATS[v, :4]) for v in ids]
        edges = [wire.Edge(src_node_id=int(rng.choice(ids)), dst_node_id=int(rng.choice(ids)), condensed_edge_type=0,
                           feature_values=rng.standard_normal(3).astype(np.float32)) for _ in range(rng.integers(0, 15))]
        samples.append(wire.SupervisedNodeClassificationSample(
            root_node=nodes[0], neighborhood=wire.Graph(nodes=nodes, edges=edges),
            root_node_labels=[wire.Label(label_type="y", label=1)]))
    nat = SupervisedNodeClassificationBatch.process_raw_pyg_samples_and_collate_fn([s.SerializeToString() for s in samples])
    ref = SupervisedNodeClassificationBatch.collate_pyg_node_classification_minibatch(samples)
    _same_graph(nat, ref)
    assert nat.graph.edge_attr.shape[1] == 3 and torch.equal(nat.graph.edge_attr, ref.graph.edge_attr)


# ---- heterogeneous samples: gigl_collate_typed_records vs traces of the reference GraphBuilder ----------------------
def _typed_records(case):
    recs = []
    for s in case["samples"]:
        nodes = [wire.Node(node_id=v, condensed_node_type=t, feature_values=np.asarray(f, np.float32))
                 for t, v, f in s["nodes"]]
        edges = [wire.Edge(src_node_id=a, dst_node_id=b, condensed_edge_type=c,
                           feature_values=np.asarray(ef, np.float32)) for c, a, b, ef in s["edges"]]
        recs.append(wire.RootedNodeNeighborhood(root_node=nodes[0],
                                                neighborhood=wire.Graph(nodes=nodes, edges=edges)).SerializeToString())
    return recs


def test_typed_collate_matches_reference_graph_builder_traces(golden_dir):
    import json

    from gigl_amd._lib import REC_ROOTED_NODE_NEIGHBORHOOD
    from gigl_amd.batches import collate_serialized_typed
    traces = json.load(open(os.path.join(golden_dir, "graph_builder_hetero_traces.json")))
    assert len(traces) >= 10
    for case in traces:
        ends = [tuple(p) for p in case["edge_type_endpoints"]]
        out = collate_serialized_typed(_typed_records(case), REC_ROOTED_NODE_NEIGHBORHOOD, case["n_node_types"], ends)
        feats = {}
        for s in case["samples"]:
            for t, v, f in s["nodes"]:
                feats[(t, v)] = np.asarray(f, np.float32)
        for t in range(case["n_node_types"]):
            g2l = {int(k): v for k, v in case["global_to_local"][t].items()}
            ids = out["node_ids"][t]
            assert len(ids) == len(g2l)
            for local, g in enumerate(ids.tolist()):  # first-seen numbering of the reference, per node type
                assert g2l[g] == local
                assert np.array_equal(out["x"][t][local], feats[(t, g)])
        for c in range(len(ends)):
            ordered = case["ordered_edges_local"][c]
            order = sorted(range(len(ordered)), key=lambda i: tuple(ordered[i]))  # coalesce(): sorted by (src, dst)
            want = np.array([ordered[i] for i in order], dtype=np.int64).reshape(-1, 2).T
            assert np.array_equal(out["edge_index"][c], want), c
            ef = case["ordered_edge_features"][c]
            if ordered and ef[0]:
                assert np.allclose(out["edge_attr"][c], np.array([ef[i] for i in order], np.float32))
            else:
                assert out["edge_attr"][c] is None
        # roots: (type, local id) of each record's root node
        for i, s in enumerate(case["samples"]):
            t, v, _ = s["nodes"][0]
            assert out["root_type"][i] == t
            assert out["root_local"][i] == case["global_to_local"][t][str(v)]


def test_typed_collate_errors():
    from gigl_amd._lib import REC_ROOTED_NODE_NEIGHBORHOOD
    from gigl_amd.batches import collate_serialized_typed
    f = np.ones(2, np.float32)
    a = wire.Node(node_id=1, condensed_node_type=0, feature_values=f)
    b = wire.Node(node_id=1, condensed_node_type=1, feature_values=f)  # same id, another type: another node
    ok = wire.RootedNodeNeighborhood(root_node=a, neighborhood=wire.Graph(
        nodes=[a, b], edges=[wire.Edge(src_node_id=1, dst_node_id=1, condensed_edge_type=0)]))
    out = collate_serialized_typed([ok.SerializeToString()], REC_ROOTED_NODE_NEIGHBORHOOD, 2, [(0, 1)])
    assert out["node_ids"][0].tolist() == [1] and out["node_ids"][1].tolist() == [1]
    assert out["edge_index"][0].tolist() == [[0], [0]]
    bad = wire.RootedNodeNeighborhood(root_node=a, neighborhood=wire.Graph(
        nodes=[a], edges=[wire.Edge(src_node_id=1, dst_node_id=1, condensed_edge_type=0)]))  # type-1 node 1 unknown
    with pytest.raises(TypeError):
        collate_serialized_typed([bad.SerializeToString()], REC_ROOTED_NODE_NEIGHBORHOOD, 2, [(0, 1)])
    other = wire.Node(node_id=1, condensed_node_type=0, feature_values=2 * f)
    clash = wire.RootedNodeNeighborhood(root_node=a, neighborhood=wire.Graph(nodes=[a, other], edges=[]))
    with pytest.raises(AssertionError):
        collate_serialized_typed([clash.SerializeToString()], REC_ROOTED_NODE_NEIGHBORHOOD, 2, [(0, 1)])
    with pytest.raises(ValueError):  # a node type outside the metadata
        collate_serialized_typed([ok.SerializeToString()], REC_ROOTED_NODE_NEIGHBORHOOD, 1, [(0, 0)])
