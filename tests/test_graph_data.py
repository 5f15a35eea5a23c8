"""host-side behaviour of nn.GraphData for batches built in HBM: a batch may leave its dense feature matrix out (x_fn) —
the node count then comes from the CSR, features() gathers once on demand, .to() of a resident batch is the identity."""
import torch

from gigl_amd.nn import GraphData


def test_graph_data_without_a_dense_feature_matrix():
    calls = []

    def x_fn():
        calls.append(1)
        return torch.arange(12, dtype=torch.float32).view(4, 3)

    g = GraphData(x=None, edge_index=torch.tensor([[1, 2, 3], [0, 0, 1]]))
    g.rowptr = torch.tensor([0, 2, 3, 3, 3], dtype=torch.int32)
    g.col = torch.tensor([1, 2, 3], dtype=torch.int32)
    g.x_fn = x_fn
    assert g.num_nodes == 4 and g.num_edges == 3
    assert g.to("cpu") is g and not calls          # (resident with its CSR: nothing is gathered by a move)
    x = g.features()
    assert tuple(x.shape) == (4, 3) and g.features() is x and len(calls) == 1
    assert g.num_nodes == 4


def test_graph_data_with_features_is_unchanged():
    g = GraphData(x=torch.ones(3, 2), edge_index=torch.tensor([[1, 2], [0, 0]]))
    assert g.num_nodes == 3 and g.features() is g.x and g.node_ids is None and g.levels is None and g.table is None
