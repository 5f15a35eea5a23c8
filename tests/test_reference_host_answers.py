"""Known answers of the reference's own unit tests for the host-side pieces of the trainer path, restated:
eval_metrics_test.py (Hits@k, MRR), early_stop_test.py (EarlyStopper), sampling-op DAG construction.  CPU only."""
import pytest
import torch
import torch.nn as nn

from gigl_amd.base import EvalMetricType, hit_rate_at_k, mean_reciprocal_rank


def test_hit_rates_known_answers():
    # python/tests/unit/src/common/utils/eval_metrics_test.py:13-55
    neg = torch.tensor([3.0, 5.0, 7.0])
    ks = torch.tensor([1, 2, 3])
    assert hit_rate_at_k(torch.tensor([9.0]), neg, ks).tolist() == [1.0, 1.0, 1.0]
    assert hit_rate_at_k(torch.tensor([6.0]), neg, ks).tolist() == [0.0, 1.0, 1.0]
    for pos in (2.0, 4.0, 6.0, 8.0):  # k = 1 + #negatives is always a hit
        assert hit_rate_at_k(torch.tensor([pos]), neg, torch.tensor([4]))[0].item() == 1.0
    far = hit_rate_at_k(torch.tensor([2.0]), neg, torch.tensor([1, 2, 3, 4, 5]))  # k beyond what the input can rank
    assert far.numel() == 5 and far[-1].item() == 1.0 and far[:3].tolist() == [0.0, 0.0, 0.0]
    with pytest.raises(AssertionError):
        hit_rate_at_k(torch.tensor([9.0]), neg, torch.tensor([0]))
    # several positives: the mean over them
    assert hit_rate_at_k(torch.tensor([9.0, 4.0]), neg, torch.tensor([1, 3])).tolist() == [0.5, 1.0]


def test_mean_reciprocal_rank_known_answers():
    # eval_metrics_test.py:57-73
    neg = torch.tensor([3.0, 5.0, 7.0])
    assert mean_reciprocal_rank(torch.tensor([9.0]), neg).item() == 1.0
    assert mean_reciprocal_rank(torch.tensor([6.0]), neg).item() == 0.5
    assert abs(mean_reciprocal_rank(torch.tensor([9.0, 2.0]), neg).item() - (1.0 + 0.25) / 2) < 1e-7


class _Dummy(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("dummy_value", torch.tensor(0.0))


@pytest.mark.parametrize("criterion", [EvalMetricType.mrr, EvalMetricType.loss])
def test_early_stopper_known_answers(criterion):
    # python/tests/unit/src/common/modeling_task_spec_utils/early_stop_test.py:21-76
    from gigl_amd.nablp_spec import EarlyStopper
    loss = [150.0, 100.0, 50.0, 60.0, 70.0, 30.0, 40.0, 50.0, 80.0]
    mrr = [0.1, 0.3, 0.5, 0.45, 0.4, 0.6, 0.5, 0.4, 0.3]
    metrics = [{EvalMetricType.loss: a, EvalMetricType.mrr: b} for a, b in zip(loss, mrr)]
    model = _Dummy()
    stopper = EarlyStopper(early_stop_criterion=criterion, early_stop_patience=3)
    for m in metrics[:-1]:
        assert not stopper.should_early_stop(metrics=m, model=model)
        model.dummy_value += 1
    assert stopper.should_early_stop(metrics=metrics[-1], model=model)  # three checks without improvement
    model.load_state_dict(stopper.best_val_model)
    assert model.dummy_value == 5  # the state at the best check (index 5)


def test_sampling_op_dag_construction():
    """SamplingOpDAG.from (scala_spark35/common/src/main/scala/types/SamplingOpDAG.scala:19-53): roots are the ops
    without inputs, children are linked by op name, an op runs after ALL its parents"""
    from gigl_amd.graphdb_sampler import EdgeType, SamplingOp, SamplingOpDAG
    et = EdgeType("a", "to", "b")
    ops = [SamplingOp("r1", et, 2), SamplingOp("r2", et, 2), SamplingOp("j", et, 2, ["r1", "r2"]),
           SamplingOp("leaf", et, 1, ["j"])]
    dag = SamplingOpDAG.from_ops(ops)
    assert dag.root_op_names == ["r1", "r2"] and dag.nodes["j"].parent_op_names == ["r1", "r2"]
    assert dag.nodes["r1"].child_op_names == ["j"] and dag.nodes["j"].child_op_names == ["leaf"]
    order = dag.execution_order()
    assert order.index("j") > max(order.index("r1"), order.index("r2")) and order[-1] == "leaf" and len(order) == 4
    with pytest.raises(ValueError):
        SamplingOpDAG.from_ops([SamplingOp("x", et, 1), SamplingOp("x", et, 1)])


def test_sampling_op_dag_validation_categories():
    """the failure categories of the reference's strategy / sampling-op validation tests (python/tests/unit/src/
    validation/subgraph_sampling_strategy_validation_test.py:190-625, sampling_op_validation_test.py:77-400) on the same
    shapes of DAG: node types '0' / '1' / '2', edge types 0->1, 1->2, 2->0"""
    from gigl_amd.graphdb_sampler import (INCOMING, OUTGOING, EdgeType, SamplingOp, SubgraphSamplingValidationError,
                                          validate_sampling_op_dags)
    nts = ["0", "1", "2"]
    e01, e12, e20 = EdgeType("0", "to", "1"), EdgeType("1", "to", "2"), EdgeType("2", "to", "0")
    ets = [e01, e12, e20]

    def fails(kind, dags, expected=()):
        with pytest.raises(SubgraphSamplingValidationError) as ei:
            validate_sampling_op_dags(dags, nts, ets, expected)
        assert ei.value.error_type == kind

    ok = {"1": [SamplingOp("a", e01, 10, [], INCOMING), SamplingOp("b", e20, 5, ["a"], INCOMING)],
          "0": [SamplingOp("a", e01, 10, [], OUTGOING), SamplingOp("b", e12, 5, ["a"], OUTGOING)],
          "2": []}  # a zero-hop DAG is valid
    validate_sampling_op_dags(ok, nts, ets, ["0", "1", "2"])
    # root ops: INCOMING needs edge_type.dst == root type, OUTGOING edge_type.src == root type
    fails("CONTAINS_INVALID_EDGE_IN_DAG", {"0": [SamplingOp("a", e01, 10, [], INCOMING)]})
    fails("CONTAINS_INVALID_EDGE_IN_DAG", {"1": [SamplingOp("a", e01, 10, [], OUTGOING)]})
    # child INCOMING after parent INCOMING: child.dst == parent.src;  after parent OUTGOING: child.dst == parent.dst
    fails("CONTAINS_INVALID_EDGE_IN_DAG", {"2": [SamplingOp("p", e12, 3, [], INCOMING), SamplingOp("c", e12, 3, ["p"], INCOMING)]})
    validate_sampling_op_dags({"2": [SamplingOp("p", e12, 3, [], INCOMING), SamplingOp("c", e01, 3, ["p"], INCOMING)]}, nts, ets)
    validate_sampling_op_dags({"1": [SamplingOp("p", e12, 3, [], OUTGOING), SamplingOp("c", e12, 3, ["p"], INCOMING)]}, nts, ets)
    fails("CONTAINS_INVALID_EDGE_IN_DAG", {"1": [SamplingOp("p", e12, 3, [], OUTGOING), SamplingOp("c", e01, 3, ["p"], INCOMING)]})
    # child OUTGOING after parent INCOMING: child.src == parent.src;  after parent OUTGOING: child.src == parent.dst
    validate_sampling_op_dags({"2": [SamplingOp("p", e12, 3, [], INCOMING), SamplingOp("c", e12, 3, ["p"], OUTGOING)]}, nts, ets)
    fails("CONTAINS_INVALID_EDGE_IN_DAG", {"2": [SamplingOp("p", e12, 3, [], INCOMING), SamplingOp("c", e20, 3, ["p"], OUTGOING)]})
    fails("REPEATED_OP_NAME", {"1": [SamplingOp("a", e01, 1, [], INCOMING), SamplingOp("a", e01, 1, [], INCOMING)]})
    fails("BAD_INPUT_OP_NAME", {"1": [SamplingOp("a", e01, 1, [], INCOMING), SamplingOp("b", e20, 1, ["nope"], INCOMING)]})
    fails("SAMPLING_OP_EDGE_TYPE_NOT_IN_GRAPH_METADATA", {"1": [SamplingOp("a", EdgeType("0", "other", "1"), 1, [], INCOMING)]})
    fails("ROOT_NODE_TYPE_NOT_IN_GRAPH_METADATA", {"9": []})
    fails("ROOT_NODE_TYPE_NOT_IN_TASK_METADATA", {"2": []}, expected=["0"])
    fails("MISSING_EXPECTED_ROOT_NODE_TYPE", {"0": []}, expected=["0", "1"])
    cyc = [SamplingOp("r", e01, 1, [], INCOMING), SamplingOp("x", e20, 1, ["r", "z"], INCOMING),
           SamplingOp("y", e12, 1, ["x"], INCOMING), SamplingOp("z", e01, 1, ["y"], INCOMING)]
    fails("DAG_CONTAINS_CYCLE", {"1": cyc})
    fails("MISSING_ROOT_SAMPLING_OP", {"1": [SamplingOp("x", e01, 1, ["y"], INCOMING), SamplingOp("y", e20, 1, ["x"], INCOMING)]})
