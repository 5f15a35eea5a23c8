"""Which link-prediction jobs the trainer hands to the library's training plans (no GPU: the predicates only).
HipNodeAnchorLinkPredictionSpec._library_train_plan picks engine.NablpTrainPlan for the plain mean-GraphSAGE encoder and
engine.GatNablpTrainPlan where GatNablpTrainPlan.applies — configs[4]'s two-layer GAT over rows wider than the first
layer's output — and keeps the autograd loop for everything else."""
import pytest


@pytest.mark.parametrize("kw,feat_dim,want", [
    (dict(in_dim=768, hid_dim=128, out_dim=128, heads=2), 768, True),        # configs[4]
    (dict(in_dim=100, hid_dim=16, out_dim=32, heads=4), 100, True),
    (dict(in_dim=100, hid_dim=16, out_dim=32, heads=1), 100, True),
    (dict(in_dim=100, hid_dim=64, out_dim=32, heads=2), 100, False),         # 2 x 64 >= 100: projecting first is cheaper
    (dict(in_dim=2, hid_dim=8, out_dim=8, heads=2), 2, False),               # the reference fixture's 2-wide rows
    (dict(in_dim=100, hid_dim=16, out_dim=32, heads=3), 100, False),         # heads outside {1, 2, 4}
    (dict(in_dim=102, hid_dim=16, out_dim=32, heads=2), 102, False),         # rows not a multiple of 4 floats
    (dict(in_dim=100, hid_dim=16, out_dim=32, heads=2, edge_dim=4), 100, False),
    (dict(in_dim=100, hid_dim=16, out_dim=32, heads=2, num_layers=3), 100, False),
    (dict(in_dim=100, hid_dim=16, out_dim=32, heads=2, activation_after_last_conv=True), 100, False),
    (dict(in_dim=100, hid_dim=16, out_dim=32, heads=2), 64, False),          # the table is not what the model expects
])
def test_gat_plan_predicate(kw, feat_dim, want):
    from gigl_amd.engine import GatNablpTrainPlan
    from gigl_amd.models_attn import GAT
    kw = dict(kw)
    model = GAT(kw.pop("in_dim"), kw.pop("hid_dim"), kw.pop("out_dim"), **kw)
    assert GatNablpTrainPlan.applies(model, feat_dim) is want


def test_other_encoders_are_not_taken_for_the_gat_plan():
    from gigl_amd.engine import GatNablpTrainPlan
    from gigl_amd.models import GraphSAGE
    from gigl_amd.models_attn import TwoLayerGCN
    assert not GatNablpTrainPlan.applies(GraphSAGE(100, 16, 8, num_layers=2), 100)
    assert not GatNablpTrainPlan.applies(TwoLayerGCN(100, 8), 100)
