"""SamplingOpDAG construction / queue order (host logic, no GPU) and the per-root restatement's own contract."""
import numpy as np

from oracle import dag_sampler
from gigl_amd.graphdb_sampler import INCOMING, OUTGOING, EdgeType, SamplingOp, SamplingOpDAG

E = EdgeType("n", "r", "n")


def test_dag_roots_children_and_queue_order():
    # SamplingOpDAG.from (SamplingOpDAG.scala:19-53): roots = ops without inputs; a child with two parents runs after
    # the later one; an op naming an unknown parent simply has fewer parents (filterKeys)
    ops = [SamplingOp("a", E, 2), SamplingOp("b", E, 2), SamplingOp("c", E, 2, ["a", "b"]), SamplingOp("d", E, 1, ["c"]),
           SamplingOp("e", E, 1, ["a", "zzz"])]
    dag = SamplingOpDAG.from_ops(ops)
    assert dag.root_op_names == ["a", "b"]
    assert dag.nodes["c"].parent_op_names == ["a", "b"] and dag.nodes["a"].child_op_names == ["c", "e"]
    assert dag.nodes["e"].parent_op_names == ["a"]
    assert dag.execution_order() == ["a", "b", "c", "e", "d"]
    # an op whose parent is never reachable never runs (incomplete graph: "path will be skipped")
    dag2 = SamplingOpDAG.from_ops([SamplingOp("x", E, 1, ["y"]), SamplingOp("y", E, 1, ["x"])])
    assert dag2.root_op_names == [] and dag2.execution_order() == []


def test_restatement_on_a_hand_graph():
    # 0 <- 1, 0 <- 2, 1 <- 3, 2 <- 3, 3 <- 4; INCOMING two hops from root 0 with fanout 2 reach {1,2} then {3}
    src = np.array([1, 2, 3, 3, 4], dtype=np.uint32)
    dst = np.array([0, 0, 1, 2, 3], dtype=np.uint32)
    nbrs = dag_sampler.neighbour_lists({E: (src, dst)})
    ops = [SamplingOp("h1", E, 2, [], INCOMING), SamplingOp("h2", E, 2, ["h1"], INCOMING)]
    edges, nodes = dag_sampler.sample_for_root(0, ops, nbrs, {"n": 0}, {E: 7}, "n")
    assert edges == {(1, 0, 7), (2, 0, 7), (3, 1, 7), (3, 2, 7)} and nodes == {(0, 0), (1, 0), (2, 0), (3, 0)}
    # OUTGOING from 3: {1, 2}, then their OUTGOING: {0}; fanout 1 keeps one of {1, 2} and its edge to 0
    ops = [SamplingOp("o1", E, 1, [], OUTGOING), SamplingOp("o2", E, 1, ["o1"], OUTGOING)]
    edges, nodes = dag_sampler.sample_for_root(3, ops, nbrs, {"n": 0}, {E: 0}, "n")
    assert len(edges) == 2 and (3, 0) in nodes and (0, 0) in nodes
    mid = ({v for v, _ in nodes} - {0, 3}).pop()
    assert edges == {(3, mid, 0), (mid, 0, 0)} and mid in (1, 2)
    # a root without neighbours: just itself; the child op never runs
    edges, nodes = dag_sampler.sample_for_root(4, [SamplingOp("h1", E, 2, [], INCOMING),
                                                   SamplingOp("h2", E, 2, ["h1"], INCOMING)], nbrs, {"n": 0}, {E: 0}, "n")
    assert edges == set() and nodes == {(4, 0)}
