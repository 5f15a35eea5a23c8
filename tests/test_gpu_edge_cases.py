"""Empty and degenerate inputs through the newer entry points (the reference's loaders meet all of these: empty
batches at the end of an epoch, isolated nodes, graphs without edges)."""
import numpy as np
import pytest
import torch

from oracle import avro
from gigl_amd import export

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from gigl_amd.engine import HipEngine
    e = HipEngine(0)
    # 6 nodes: 0 <- 1, 0 <- 2, 3 isolated, 4 <-> 5
    e.build_from_coo(6, np.array([1, 2, 4], np.uint32), np.array([0, 0, 5], np.uint32), is_directed=False)
    e.load_features(np.arange(12, dtype=np.float32).reshape(6, 2))
    yield e
    e.close()


def test_empty_embedding_batch_and_exporter(eng, tmp_path):
    blocks, off = eng.encode_avro_embeddings(torch.zeros(0, dtype=torch.int64), torch.zeros((0, 4)), "t", bytes(16))
    assert blocks.numel() == 0 and off.numel() == 0
    with export.EmbeddingExporter(str(tmp_path), engine=eng) as ex:
        ex.add_embedding(torch.zeros(0, dtype=torch.int64), torch.zeros((0, 4)), "t")   # header only
        ex.add_embedding(torch.tensor([7]), torch.tensor([[1.5, -2.0, 0.0, 3.0]]), "t")
    schema, recs = avro.read_embedding_file((tmp_path / "shard_00000000.avro").read_bytes())
    assert recs == [{"node_id": 7, "node_type": "t", "emb": [1.5, -2.0, 0.0, 3.0]}]
    with pytest.raises(ValueError):
        export.EmbeddingExporter(str(tmp_path), engine=eng).add_embedding(torch.tensor([1, 2]), torch.zeros((3, 2)), "t")
    with pytest.raises(TypeError):
        export.EmbeddingExporter(str(tmp_path), engine=eng).add_embedding(torch.tensor([1.0]), torch.zeros((1, 2)), "t")


def test_edge_ids_degenerate(eng):
    e = torch.zeros(0, dtype=torch.int32)
    assert eng.edge_ids(e, e).numel() == 0
    src = torch.tensor([1, 0, 3, 5, 4, 2, 2], dtype=torch.int32)
    dst = torch.tensor([0, 1, 3, 4, 5, 0, 1], dtype=torch.int32)
    got = eng.edge_ids(src, dst).cpu().tolist()
    rowptr, col = eng.graph_to_host()
    assert got[2] == -1 and got[6] == -1  # isolated node / not an edge
    for k in (0, 1, 3, 4, 5):
        assert col[got[k]] == int(src[k]) and rowptr[int(dst[k])] <= got[k] < rowptr[int(dst[k]) + 1]


def test_records_for_isolated_and_edgeless(eng):
    from gigl_amd import wire
    eng.load_edge_features(np.array([1, 2, 4], np.uint32), np.array([0, 0, 5], np.uint32),
                           np.array([[1.0], [2.0], [3.0]], np.float32), is_directed=False)
    tree = eng.sample_khop(np.array([3, 0, 5], np.uint32), [2, 2])
    buf, off = eng.encode_records(tree)
    recs = [wire.RootedNodeNeighborhood.FromString(r) for r in wire.iter_tfrecords(buf.cpu().numpy().tobytes())]
    assert [r.root_node.node_id for r in recs] == [3, 0, 5]
    assert [n.node_id for n in recs[0].neighborhood.nodes] == [3] and recs[0].neighborhood.edges == []
    feats = {(e.src_node_id, e.dst_node_id): float(e.feature_values[0]) for e in recs[1].neighborhood.edges}
    assert feats[(1, 0)] == 1.0 and feats[(2, 0)] == 2.0 and feats.get((0, 1), 1.0) == 1.0
    assert {(e.src_node_id, e.dst_node_id) for e in recs[2].neighborhood.edges} == {(4, 5), (5, 4)}
    assert all(float(e.feature_values[0]) == 3.0 for e in recs[2].neighborhood.edges)


def test_dag_sampler_on_a_graph_with_dead_ends():
    from gigl_amd.graphdb_sampler import INCOMING, OUTGOING, EdgeType, HipGraphDBSampler, SamplingOp, SamplingOpDAG
    et = EdgeType("n", "r", "n")
    s = HipGraphDBSampler({"n": 0}, {"n": 6}, {et: (np.array([1, 2, 4], np.uint32), np.array([0, 0, 5], np.uint32))},
                          {et: 0})
    dag = SamplingOpDAG.from_ops([SamplingOp("a", et, 2, [], INCOMING), SamplingOp("b", et, 2, ["a"], INCOMING),
                                  SamplingOp("c", et, 1, ["a", "b"], OUTGOING)])
    msgs = s.getKHopSubgraphForRootNodes([0, 3, 5, 4], "n", dag)
    as_sets = [({(e.src_node_id, e.dst_node_id) for e in m.neighborhood.edges}, {n.node_id for n in m.neighborhood.nodes})
               for m in msgs]
    # root 0: a = {1, 2}; b: 1 and 2 have no in-edges -> returns nothing (but ran); c: frontier {1, 2} -> out-edges to 0
    assert as_sets[0] == ({(1, 0), (2, 0)}, {0, 1, 2})
    assert as_sets[1] == (set(), {3})                 # isolated root: a returns nothing, b and c never run
    assert as_sets[2] == ({(4, 5)}, {4, 5})           # a = {4}; b: 4 has no in-edge; c: 4 -> 5
    assert as_sets[3] == (set(), {4})
    s.close()
