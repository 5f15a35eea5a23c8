"""The library's training step (gigl_sage_train_plan_*, csrc/pipeline.hip) against the autograd path it replaces and
against the CPU restatement of the reference's loop
(/root/reference/python/gigl/src/common/modeling_task_specs/node_classification_modeling_task_spec.py:134-173:
zero_grad -> model(inputs) -> F.cross_entropy(out[root_node_indices], labels) -> backward -> Adam(lr 0.01, wd 5e-4)):
the same batches from the same initial weights give the same loss history (1e-6 relative per step against the autograd
path on the device, 1e-4 against fp32 CPU autograd over oracle-collated batches) and the same trained weights."""
import numpy as np
import pytest
import torch

import oracle
from helpers import rmat_edges
from oracle import gnn_ref


@pytest.fixture(scope="module")
def setup():
    from gigl_amd.engine import HipEngine
    s, d = rmat_edges(13, 150000, seed=8)
    n = 1 << 13
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    x = (np.random.default_rng(0).standard_normal((n, 100)) / 4).astype(np.float32)
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    yield eng, rowptr, col, x, n
    eng.close()


def _autograd_losses(eng, model, roots_all, labels_all, b, fan, steps, lr, wd):
    """the step GraphedTrainStep captures, eagerly: sample + union in HBM, forward with autograd, CE, backward, Adam"""
    import torch.nn.functional as F
    from gigl_amd.models import HipBatch
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
    losses = []
    for i in range(steps):
        roots = roots_all[i * b:(i + 1) * b]
        tree = eng.sample_khop(roots, fan)
        u = eng.union_build(tree)
        out = model(HipBatch(eng, tree, u, train=True))
        loss = F.cross_entropy(out[u.root_local[: roots.numel()].long()], labels_all[i * b:(i + 1) * b])
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses


@pytest.mark.gpu
@pytest.mark.parametrize("dims,fan,b,prefetch", [((100, 64, 7), [10, 5], 256, False), ((100, 256, 47), [25, 10], 128, True),
                                                  ((100, 64, 7), [10, 5], 256, True)])
def test_library_training_step_equals_the_autograd_step(setup, dims, fan, b, prefetch):
    from gigl_amd.engine import SageTrainPlan
    from gigl_amd.models import GraphSAGE
    eng, rowptr, col, x, n = setup
    steps = 12
    rng = np.random.default_rng(3)
    roots_all = torch.from_numpy(rng.permutation(n)[: steps * b].astype(np.uint32).view(np.int32)).to(eng.device)
    labels_all = torch.from_numpy(rng.integers(0, dims[2], steps * b)).to(eng.device)
    torch.manual_seed(1)
    ref = GraphSAGE(dims[0], dims[1], dims[2], num_layers=2).to(eng.device)
    lib = GraphSAGE(dims[0], dims[1], dims[2], num_layers=2).to(eng.device)
    lib.load_state_dict(ref.state_dict())
    ref.train()
    want = _autograd_losses(eng, ref, roots_all, labels_all, b, fan, steps, 0.01, 5e-4)
    st = torch.cuda.Stream(device=eng.device)
    torch.cuda.synchronize()
    eng.bind_stream(st)
    plan = SageTrainPlan(eng, lib, b, fan, lr=0.01, weight_decay=5e-4)
    got = []
    with torch.cuda.stream(st):  # (the loss is a device scalar written on the plan's stream)
        for i in range(steps):  # (the parts run eagerly once, are captured on their second run, then replayed)
            # every other step hands the plan the next batch's roots: its graph part then overlaps this step's layers
            nxt = roots_all[(i + 1) * b:(i + 2) * b] if (prefetch and i + 1 < steps and i % 3 != 2) else None
            got.append(plan.step(roots_all[i * b:(i + 1) * b], labels_all[i * b:(i + 1) * b], next_roots=nxt).clone())
    eng.synchronize()
    got = [float(v) for v in got]
    plan.store(lib)
    plan.close()
    eng.bind_stream(torch.cuda.current_stream(eng.device))
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
    for (k, a), (_, bb) in zip(lib.state_dict().items(), ref.state_dict().items()):
        # (Adam divides by sqrt(v): an element whose gradients are rounding noise moves by up to lr per step either way)
        np.testing.assert_allclose(a.cpu().numpy(), bb.cpu().numpy(), rtol=2e-3, atol=2e-4, err_msg=k)


@pytest.mark.gpu
def test_library_training_step_against_the_cpu_restatement(setup):
    """oracle sample -> collate -> fp32 CPU autograd -> Adam, five steps; a short last batch is padded and masked"""
    from gigl_amd.engine import SageTrainPlan
    from gigl_amd.models import GraphSAGE
    eng, rowptr, col, x, n = setup
    b, fan, steps = 64, [10, 5], 5
    rng = np.random.default_rng(9)
    roots_np = rng.permutation(n)[: steps * b - 20].astype(np.uint32)  # (the last batch has 44 real roots)
    labels_np = rng.integers(0, 7, roots_np.size)
    torch.manual_seed(2)
    model = GraphSAGE(100, 32, 7, num_layers=2)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(list(params.values()), lr=0.01, weight_decay=5e-4)
    want = []
    for lo in range(0, roots_np.size, b):
        roots = roots_np[lo:lo + b]
        nbr, _ = oracle.sample_khop(rowptr, col, roots, fan, canonical=True)
        u = oracle.union_build(roots, fan, nbr)
        ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
        out = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"].astype(np.int64)]), ei, params, 2)
        loss = torch.nn.functional.cross_entropy(out[torch.from_numpy(u["root_local"].astype(np.int64))],
                                                 torch.from_numpy(labels_np[lo:lo + b]))
        opt.zero_grad()
        loss.backward()
        opt.step()
        want.append(float(loss.detach()))
    model = model.to(eng.device)
    st = torch.cuda.Stream(device=eng.device)
    torch.cuda.synchronize()
    eng.bind_stream(st)
    plan = SageTrainPlan(eng, model, b, fan, lr=0.01, weight_decay=5e-4)
    r_dev = torch.from_numpy(roots_np.view(np.int32)).to(eng.device)
    l_dev = torch.from_numpy(labels_np).to(eng.device)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        got = [plan.step(r_dev[lo:lo + b], l_dev[lo:lo + b]).clone() for lo in range(0, roots_np.size, b)]
    eng.synchronize()
    got = [float(v) for v in got]
    plan.store(model)
    plan.close()
    eng.bind_stream(torch.cuda.current_stream(eng.device))
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), params[k].detach().numpy(), rtol=1e-3, atol=1e-4, err_msg=k)
