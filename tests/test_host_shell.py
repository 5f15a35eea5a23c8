"""Host-side mirror of the reference interfaces (no GPU): config reader, sample assembly, collate, metrics,
TFRecord sharding, and loud failure of the entry points without a device."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from gigl_amd import wire
from gigl_amd.base import hit_rate_at_k, mean_reciprocal_rank
from gigl_amd.batches import (RootedNodeNeighborhoodBatch, SupervisedNodeClassificationBatch, build_batch_graph,
                              iterate_tfrecord_batches)
from gigl_amd.config import GbmlConfigPbWrapper, tfrecord_files
from gigl_amd.sampler_service import (build_rooted_node_neighborhood, tree_to_edge_lists,
                                      validate_rooted_node_neighborhood)
from gigl_amd.subgraph_sampler import load_preprocessed_graph
from helpers import A, check_rnn_validity, load_fixture_graph


def test_config_reader(golden_dir):
    cfg = GbmlConfigPbWrapper.from_uri("configs/snc_frozen_gbml_config.yaml", uri_base=golden_dir)
    assert cfg.task_kind == "node_classification" and cfg.fanouts == [3, 3] and not cfg.is_graph_directed
    assert cfg.num_positive_samples == 2 and cfg.permutation_strategy == "deterministic"
    assert cfg.trainer_cls_path.endswith("HipGraphSageNodeClassificationSpec")
    assert cfg.trainer_args["num_epochs"] == "3"  # plugin kwargs stay strings, like the proto map<string,string>
    assert cfg.inference_batch_size == 8
    pm = cfg.preprocessed_metadata
    assert pm.nodes[0].feature_keys == ["f0", "f1"] and pm.nodes[0].label_keys == ["node_label"]
    assert pm.edges[0].src_node_id_key == "src"
    cfg2 = GbmlConfigPbWrapper.from_uri("configs/nablp_frozen_gbml_config.yaml", uri_base=golden_dir)
    assert cfg2.task_kind == "node_anchor_based_link_prediction"
    assert list(cfg2.random_negative_tfrecord_uri_prefixes) == ["user"]
    with pytest.raises(NotImplementedError):
        GbmlConfigPbWrapper({"sharedConfig": {"preprocessedMetadataUri": "gs://bucket/x.yaml"}}).preprocessed_metadata


def test_per_hop_fanouts_from_sampling_ops():
    doc = {"datasetConfig": {"subgraphSamplerConfig": {"subgraphSamplingStrategy": {"messagePassingPaths": {"paths": [
        {"rootNodeType": "user", "samplingOps": [
            {"opName": "hop1", "randomUniform": {"numNodesToSample": 25}},
            {"opName": "hop2", "inputOpNames": ["hop1"], "randomUniform": {"numNodesToSample": 10}}]}]}}}}}
    assert GbmlConfigPbWrapper(doc).fanouts == [25, 10]


def test_ingest_matches_fixture_loader(golden_dir):
    cfg = GbmlConfigPbWrapper.from_uri("configs/snc_frozen_gbml_config.yaml", uri_base=golden_dir)
    n, src, dst, x, labels, ids = load_preprocessed_graph(cfg)
    n2, src2, dst2, feats2 = load_fixture_graph(golden_dir)
    assert n == n2 == 16 and np.array_equal(np.sort(src * 100 + dst), np.sort(src2 * 100 + dst2))
    assert np.allclose(x, feats2) and len(labels["node_label"]) == 16 and ids == list(range(16))


def test_rnn_assembly_from_trees(golden_dir):
    """tree layout -> RootedNodeNeighborhood: reference node order (hop-1 ++ hop-2 ++ root, distinct), validity,
    TaskOutputValidator, and a byte round trip"""
    n, src, dst, feats = load_fixture_graph(golden_dir)
    rowptr, col = oracle.build_csc(n, src, dst, is_directed=False)
    roots = np.arange(n, dtype=np.uint32)
    fan = [3, 3]
    nbr, cnt = oracle.sample_khop(rowptr, col, roots, fan, canonical=True)
    lists = tree_to_edge_lists(roots, fan, nbr)
    for r, (s, d) in zip(roots.tolist(), lists):
        rnn = build_rooted_node_neighborhood(r, s, d, feats)
        validate_rooted_node_neighborhood(rnn)
        edges = [(e.src_node_id, e.dst_node_id) for e in rnn.neighborhood.edges]
        nodes = [x.node_id for x in rnn.neighborhood.nodes]
        check_rnn_validity(r, edges, nodes, rowptr, col, fanout=3)
        assert len(edges) == int(cnt[0][r] + cnt[1][r * 3:(r + 1) * 3].sum())
        if edges:
            assert nodes[-1] == r or r in [e[0] for e in edges]  # root appended last unless already a source
        for x in rnn.neighborhood.nodes:
            assert np.allclose(x.feature_values, feats[x.node_id]) and x.condensed_node_type == 0
        assert wire.RootedNodeNeighborhood.FromString(rnn.SerializeToString()).SerializeToString() == rnn.SerializeToString()
    iso = [r for r, (s, d) in zip(roots.tolist(), lists) if s.size == 0]
    assert iso == [14, 15]
    bad = wire.RootedNodeNeighborhood(root_node=wire.Node(node_id=1), neighborhood=wire.Graph(
        nodes=[wire.Node(node_id=1)], edges=[wire.Edge(src_node_id=2, dst_node_id=1)]))
    with pytest.raises(RuntimeError):
        validate_rooted_node_neighborhood(bad)


def _n(i):
    return wire.Node(node_id=i, condensed_node_type=0, feature_values=np.array([float(i), 1.0], np.float32))


def _e(s, d):
    return wire.Edge(src_node_id=s, dst_node_id=d, condensed_edge_type=0)


def test_collate_known_answers_and_traces(golden_dir):
    """rooted_node_neighborhood_batching_test.py:150-210 restated + the reference GraphBuilder traces"""
    tri = wire.RootedNodeNeighborhood(root_node=_n(0), neighborhood=wire.Graph(
        nodes=[_n(0), _n(1), _n(2)], edges=[_e(0, 1), _e(0, 2), _e(1, 2)]))
    line = wire.RootedNodeNeighborhood(root_node=_n(3), neighborhood=wire.Graph(nodes=[_n(3), _n(4)], edges=[_e(3, 4)]))
    chain = wire.RootedNodeNeighborhood(root_node=_n(2), neighborhood=wire.Graph(
        nodes=[_n(1), _n(2), _n(3)], edges=[_e(1, 2), _e(2, 3)]))
    b = RootedNodeNeighborhoodBatch.collate_pyg_rooted_node_neighborhood_minibatch([tri, line])
    assert (b.graph.num_nodes, b.graph.num_edges) == (5, 4)
    assert b.condensed_node_type_to_root_node_indices_map[0].tolist() == [0, 3]
    b = RootedNodeNeighborhoodBatch.collate_pyg_rooted_node_neighborhood_minibatch([tri, chain])
    assert (b.graph.num_nodes, b.graph.num_edges) == (4, 4)
    assert [r.id for r in b.root_nodes] == [0, 2]
    assert b.condensed_node_type_to_subgraph_id_to_global_node_id[0] == {0: 0, 1: 1, 2: 2, 3: 3}
    # raw bytes entry point == proto entry point
    b2 = RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn([tri.SerializeToString(), chain.SerializeToString()])
    assert torch.equal(b2.graph.edge_index, b.graph.edge_index) and torch.equal(b2.graph.x, b.graph.x)
    # empty-edge batch: edge_index (2, 0)   (data_loaders/utils.py:134-145)
    lone = wire.RootedNodeNeighborhood(root_node=_n(9), neighborhood=wire.Graph(nodes=[_n(9)]))
    b3 = RootedNodeNeighborhoodBatch.collate_pyg_rooted_node_neighborhood_minibatch([lone])
    assert tuple(b3.graph.edge_index.shape) == (2, 0) and b3.graph.num_nodes == 1
    with pytest.raises(TypeError):
        build_batch_graph([([_n(1)], [_e(1, 5)])])
    # the reference builder's own traces (generated by importing abstract_graph_builder.py)
    for t in json.load(open(os.path.join(golden_dir, "graph_builder_traces.json"))):
        samples = []
        for s in t["samples"]:
            seen, es = set(), []
            for a, c in s["edges"]:
                if (a, c) not in seen:
                    seen.add((a, c)); es.append(_e(a, c))
            samples.append(([_n(v) for v in s["nodes"]], es))
        x, ei, g2l, order, _ = build_batch_graph(samples)
        assert g2l == {int(k): v for k, v in t["global_to_local"].items()}
        want = sorted(map(tuple, t["ordered_edges_local"]))  # coalesce(): sorted by (src, dst)
        assert [tuple(p) for p in ei.T.tolist()] == want
    # same answer as the oracle's C restatement
    nodes, ls, ld = oracle.collate_reference([np.array([0, 1, 2]), np.array([1, 2, 3])],
                                             [(np.array([0, 0, 1]), np.array([1, 2, 2])), (np.array([1, 2]), np.array([2, 3]))])
    assert nodes.tolist() == [0, 1, 2, 3] and list(zip(ls.tolist(), ld.tolist())) == [tuple(p) for p in b.graph.edge_index.T.tolist()]


def test_snc_collate_labels():
    s1 = wire.SupervisedNodeClassificationSample(root_node=_n(0), neighborhood=wire.Graph(nodes=[_n(0), _n(1)], edges=[_e(1, 0)]),
                                                 root_node_labels=[wire.Label(label_type="node_label", label=2)])
    s2 = wire.SupervisedNodeClassificationSample(root_node=_n(1), neighborhood=wire.Graph(nodes=[_n(1), _n(0)], edges=[_e(0, 1)]),
                                                 root_node_labels=[wire.Label(label_type="node_label", label=0)])
    b = SupervisedNodeClassificationBatch.process_raw_pyg_samples_and_collate_fn([s1.SerializeToString(), s2.SerializeToString()])
    assert b.root_node_indices.tolist() == [0, 1] and b.root_node_labels.tolist() == [2, 0]
    assert b.graph.num_nodes == 2 and b.graph.num_edges == 2


def test_eval_metrics_against_reference_outputs(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "eval_metrics.json"))):
        pos, neg, ks = torch.tensor(c["pos"]), torch.tensor(c["neg"]), torch.tensor(c["ks"])
        assert np.allclose(hit_rate_at_k(pos, neg, ks).tolist(), c["hits"], atol=1e-6)
        assert abs(float(mean_reciprocal_rank(pos, neg)) - c["mrr"]) < 1e-6


def test_tfrecord_batch_iteration_and_rank_sharding(golden_dir, tmp_path):
    recs = list(wire.read_tfrecords(os.path.join(golden_dir, A, "split_generator/supervised_node_classification/sgs_output/unlabeled/samples/data.tfrecord")))
    for i in range(4):
        wire.write_tfrecords(str(tmp_path / f"part-{i:05d}.tfrecord"), recs[i * 4:(i + 1) * 4])
    files = tfrecord_files(str(tmp_path) + "/")
    assert len(files) == 4
    got = [r for b in iterate_tfrecord_batches(files, 3) for r in b]
    assert sorted(got) == sorted(recs)
    r0 = [r for b in iterate_tfrecord_batches(files, 5, rank=0, world_size=2) for r in b]
    r1 = [r for b in iterate_tfrecord_batches(files, 5, rank=1, world_size=2) for r in b]
    assert len(r0) == len(r1) == 8 and sorted(r0 + r1) == sorted(recs)  # files strided across ranks
    it = iterate_tfrecord_batches(files[:1], 3, loop=True)
    assert sum(len(next(it)) for _ in range(5)) > 4  # loopy dataset keeps going
    # more ranks than files: the list is tiled (get_data_split_for_current_worker, data_loaders/utils.py:38-47) —
    # every rank gets batches (a rank without any would leave the others waiting in the DDP all-reduce)
    for world in (2, 3, 5):
        per_rank = [[r for b in iterate_tfrecord_batches(files[:1], 3, rank=k, world_size=world) for r in b]
                    for k in range(world)]
        assert all(len(x) == 4 for x in per_rank)
    per_rank = [[r for b in iterate_tfrecord_batches(files[:3], 3, rank=k, world_size=4) for r in b] for k in range(4)]
    assert all(len(x) >= 4 for x in per_rank)


def test_entry_points_fail_loudly_without_gpu(golden_dir):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd.trainer import Trainer
    cfg = "configs/snc_frozen_gbml_config.yaml"
    with pytest.raises(RuntimeError):
        SubgraphSampler().run("job", cfg, None, uri_base=golden_dir)
    with pytest.raises(RuntimeError):
        Trainer().run("job", cfg, None, uri_base=golden_dir)
    with pytest.raises(RuntimeError):
        Inferencer().run("job", cfg, None, uri_base=golden_dir)


def test_sampler_call_size_follows_the_record_size():
    """roots per library call of the sampler job: up to 32,768 (where the device encoder reaches 0.35 of the HBM roofline), fewer
    when a call's frames would pass the byte budget, never more than there are roots"""
    from gigl_amd.subgraph_sampler import sampler_call_size
    assert sampler_call_size(1 << 30, [25, 10], 100) == 32768           # products-shaped: 276 nodes x ~0.4 KB
    assert sampler_call_size(1 << 30, [25, 10], 768) == 4096            # MAG240M-wide rows: 4 GiB / ~0.9 MB per record
    assert sampler_call_size(1 << 30, [25, 10], 768, budget_bytes=32 << 30) == 32768
    assert sampler_call_size(2708, [10, 5], 1433) == 2708               # Cora: one call
    assert sampler_call_size(1 << 30, [1000, 1000], 1024) == 1024       # (never below 1,024 while there are that many roots)
    assert sampler_call_size(0, [10, 5], 8) == 1


def test_adam_state_errors_separate_noise_from_determined_elements():
    """tests/helpers.adam_state_errors: an element whose gradient is rounding noise may sit anywhere within lr * steps of the
    reference and must not fail the comparison; a determined element that is off must"""
    import torch
    from helpers import adam_state_errors
    g = torch.tensor([1.0, 0.5, 1e-9, -2e-9])           # two real gradients, two noise ones
    ref_p, ref_m, ref_v = torch.tensor([0.1, 0.2, 0.3, 0.4]), 0.1 * g, 0.001 * g * g
    got_p = ref_p + torch.tensor([1e-7, -1e-7, 5e-3, -5e-3])   # the noise elements drifted by lr each
    errs = adam_state_errors({"w": got_p}, {"w": (ref_m.clone(), ref_v.clone())}, {"w": ref_p}, {"w": (ref_m, ref_v)})
    em, ev, ep, share = errs["w"]
    assert em == 0 and ev == 0 and ep <= 2e-7 and abs(share - 0.5) < 1e-9
    bad = adam_state_errors({"w": ref_p + torch.tensor([1e-3, 0, 0, 0])}, {"w": (ref_m, ref_v)}, {"w": ref_p}, {"w": (ref_m, ref_v)})
    assert bad["w"][2] > 5e-4
