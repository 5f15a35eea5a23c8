"""NodeAnchorBasedLinkPredictionBatch collate — restates the homogeneous cases of the reference's
python/tests/unit/src/training/lib/data_loaders/node_anchor_based_link_prediction_batching_test.py
(:388-443 without edge overlap, :445-500 with edge overlap), plus the host pieces of the task spec
(early stopping, task container, kwargs) that need no GPU."""
import pytest
import torch

from gigl_amd import wire
from gigl_amd.batches import NodeAnchorBasedLinkPredictionBatch


def _node(i):
    return wire.Node(node_id=i, condensed_node_type=0, feature_values=[0.0])


def _edge(s, d):
    return wire.Edge(src_node_id=s, dst_node_id=d, condensed_edge_type=0)


def _samples():
    n = [_node(i) for i in range(5)]
    triangle = wire.NodeAnchorBasedLinkPredictionSample(
        root_node=n[0], pos_edges=[_edge(0, 1)], hard_neg_edges=[_edge(0, 3)],
        neighborhood=wire.Graph(nodes=[n[0], n[1], n[2], n[3]], edges=[_edge(0, 1), _edge(0, 2), _edge(1, 2)]))
    line = wire.NodeAnchorBasedLinkPredictionSample(
        root_node=n[3], pos_edges=[_edge(3, 4)], hard_neg_edges=[_edge(3, 0)],
        neighborhood=wire.Graph(nodes=[n[3], n[4], n[0]], edges=[_edge(3, 4)]))
    chain = wire.NodeAnchorBasedLinkPredictionSample(
        root_node=n[2], pos_edges=[_edge(2, 3)], hard_neg_edges=[_edge(2, 4)],
        neighborhood=wire.Graph(nodes=[n[1], n[2], n[3], n[4]], edges=[_edge(1, 2), _edge(2, 3)]))
    return triangle, line, chain


def _check_supervision(batch, pos_edges, neg_edges):
    l2g = batch.condensed_node_type_to_subgraph_id_to_global_node_id[0]
    g2l = {g: l for l, g in l2g.items()}
    pos = batch.pos_supervision_edge_data[0].root_node_to_target_node_id
    neg = batch.hard_neg_supervision_edge_data[0].root_node_to_target_node_id
    for s, d in pos_edges:
        assert g2l[s] in pos and g2l[d] in pos[g2l[s]].tolist()
    for s, d in neg_edges:
        assert g2l[s] in neg and g2l[d] in neg[g2l[s]].tolist()
    assert batch.pos_supervision_edge_data[0].label_edge_features is None
    assert batch.hard_neg_supervision_edge_data[0].label_edge_features is None


def test_collate_without_edge_overlap():
    triangle, line, _ = _samples()
    # round-trip through the wire format, as the data loader does
    raw = [triangle.SerializeToString(), line.SerializeToString()]
    batch = NodeAnchorBasedLinkPredictionBatch.process_raw_pyg_samples_and_collate_fn(raw)
    assert batch.graph.x.shape[0] == 5 and batch.graph.edge_index.shape[1] == 4
    _check_supervision(batch, [(0, 1), (3, 4)], [(0, 3), (3, 0)])
    l2g = batch.condensed_node_type_to_subgraph_id_to_global_node_id[0]
    assert [l2g[i] for i in batch.root_node_indices.tolist()] == [0, 3]


def test_collate_with_edge_overlap():
    triangle, _, chain = _samples()
    batch = NodeAnchorBasedLinkPredictionBatch.collate_pyg_node_anchor_based_link_prediction_minibatch(
        [triangle, chain])
    assert batch.graph.x.shape[0] == 5 and batch.graph.edge_index.shape[1] == 4  # 1->2 is not duplicated
    _check_supervision(batch, [(0, 1), (2, 3)], [(0, 3), (2, 4)])
    l2g = batch.condensed_node_type_to_subgraph_id_to_global_node_id[0]
    got = sorted((l2g[s], l2g[d]) for s, d in batch.graph.edge_index.t().tolist())
    assert got == [(0, 1), (0, 2), (1, 2), (2, 3)]


def test_supervision_target_outside_neighborhood_is_an_error():
    n0, n1 = _node(0), _node(1)
    bad = wire.NodeAnchorBasedLinkPredictionSample(root_node=n0, pos_edges=[_edge(0, 9)],
                                                   neighborhood=wire.Graph(nodes=[n0, n1], edges=[_edge(0, 1)]))
    with pytest.raises(KeyError):
        NodeAnchorBasedLinkPredictionBatch.collate_pyg_node_anchor_based_link_prediction_minibatch([bad])


def test_early_stopper_and_spec_kwargs():
    from gigl_amd.base import EvalMetricType
    from gigl_amd.nablp_spec import EarlyStopper, HipNodeAnchorLinkPredictionSpec
    with pytest.raises(NotImplementedError):
        EarlyStopper(EvalMetricType.hits, 3)
    lin = torch.nn.Linear(2, 2)
    es = EarlyStopper(EvalMetricType.loss, 2)
    assert not es.should_early_stop({EvalMetricType.loss: 1.0}, lin) and es.prev_best == 1.0
    best = {k: v.clone() for k, v in es.best_val_model.items()}
    with torch.no_grad():
        lin.weight.add_(1.0)
    assert not es.should_early_stop({EvalMetricType.loss: 1.5}, lin)
    assert es.should_early_stop({EvalMetricType.loss: 1.2}, lin)  # patience 2 reached
    assert all(torch.equal(best[k], es.best_val_model[k]) for k in best)  # snapshot of the best, not the latest
    mx = EarlyStopper(EvalMetricType.mrr, 1)
    assert not mx.should_early_stop({EvalMetricType.mrr: 0.2}, lin)
    assert mx.should_early_stop({EvalMetricType.mrr: 0.1}, lin)
    # string-valued kwargs, as they arrive from trainerArgs
    spec = HipNodeAnchorLinkPredictionSpec(hidden_dim="8", out_channels="4", main_sample_batch_size="3",
                                           should_remove_accidental_hits="False", softmax_temp="0.5")
    assert spec.hidden_dim == 8 and spec.out_channels == 4 and spec.main_sample_batch_size == 3
    task = spec.tasks._task_to_fn_map["Retrieval"]
    assert task.loss._temperature == 0.5 and task.loss._remove_accidental_hits is False
    assert spec.supports_distributed_training
    with pytest.raises(ValueError):
        spec.gbml_config_pb_wrapper
    with pytest.raises(RuntimeError):  # no CPU fallback
        spec._model = torch.nn.Linear(1, 1)
        spec._ensure_engine(torch.device("cpu"))
