"""Margin / Softmax link-prediction tasks (task.py:62-105, loss.py:21-174) on per-root score lists == the per-sample
formulas written with torch's own margin_ranking_loss / cross_entropy, values and gradients.  CPU only (the tasks are
small tensor algebra on scores the HIP decoder produced)."""
import torch
import torch.nn.functional as F

from gigl_amd.nablp_spec import BatchScores, Margin, NodeAnchorBasedLinkPredictionTaskInputs, Softmax


def _inputs(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(3, 2, 5), (1, 0, 5), (0, 4, 5), (2, 3, 5)]  # (positives, hard negatives, random negatives) per root
    leaves, batch = [], []
    for p, h, r in shapes:
        ts = [torch.randn(1, k, generator=g, requires_grad=True) if k else torch.zeros((0,)) for k in (p, h, r)]
        leaves += [t for t in ts if t.requires_grad]
        batch.append({0: BatchScores(pos_scores=ts[0], hard_neg_scores=ts[1], random_neg_scores=ts[2])})
    ti = NodeAnchorBasedLinkPredictionTaskInputs(main_batch=None, random_neg_batch=None, batch_embeddings=None,
                                                 batch_scores=batch)
    return ti, leaves, batch


def _reference_margin(batch, margin):
    total, n = torch.zeros(()), 0
    for result in batch:
        for bs in result.values():
            if not bs.pos_scores.numel():
                continue
            negs = torch.cat((bs.hard_neg_scores.reshape(1, -1), bs.random_neg_scores.reshape(1, -1)), dim=1)
            neg_rep = negs.repeat(1, bs.pos_scores.shape[1])
            pos_rep = bs.pos_scores.repeat_interleave(negs.shape[1], dim=1)
            total = total + F.margin_ranking_loss(pos_rep, neg_rep, torch.ones_like(pos_rep), margin=margin, reduction="sum")
            n += pos_rep.numel()
    return total, n


def _reference_softmax(batch, t):
    total, n = torch.zeros(()), 0
    for result in batch:
        for bs in result.values():
            if not bs.pos_scores.numel():
                continue
            negs = torch.cat((bs.hard_neg_scores.reshape(-1), bs.random_neg_scores.reshape(-1)))
            rows = torch.cat((bs.pos_scores.reshape(-1, 1), negs.repeat(bs.pos_scores.shape[1], 1)), dim=1)
            total = total + F.cross_entropy(rows / t, torch.zeros(rows.shape[0], dtype=torch.long), reduction="sum")
            n += bs.pos_scores.shape[1]
    return total, n


def test_margin_task_value_and_gradients():
    ti, leaves, batch = _inputs(1)
    loss, n = Margin(margin=0.5)(ti, None, False, torch.device("cpu"))
    want, n_want = _reference_margin(batch, 0.5)
    assert n == n_want == 3 * 7 + 1 * 5 + 2 * 8 and torch.allclose(loss, want, atol=1e-6)
    got = torch.autograd.grad(loss, leaves, allow_unused=True)
    ref = torch.autograd.grad(want, leaves, allow_unused=True)
    for a, b in zip(got, ref):
        assert (a is None and b is None) or torch.allclose(a if a is not None else torch.zeros_like(b),
                                                           b if b is not None else torch.zeros_like(a), atol=1e-6)


def test_softmax_task_value_and_gradients():
    ti, leaves, batch = _inputs(2)
    loss, n = Softmax(softmax_temperature=0.07)(ti, None, False, torch.device("cpu"))
    want, n_want = _reference_softmax(batch, 0.07)
    assert n == n_want == 6 and torch.allclose(loss, want, rtol=1e-5, atol=1e-5)
    got = torch.autograd.grad(loss, leaves, allow_unused=True)
    ref = torch.autograd.grad(want, leaves, allow_unused=True)
    for a, b in zip(got, ref):
        assert (a is None and b is None) or torch.allclose(a if a is not None else torch.zeros_like(b),
                                                           b if b is not None else torch.zeros_like(a), rtol=1e-4, atol=1e-5)


def test_tasks_without_any_positive():
    empty = torch.zeros((0,))
    ti = NodeAnchorBasedLinkPredictionTaskInputs(
        main_batch=None, random_neg_batch=None, batch_embeddings=None,
        batch_scores=[{0: BatchScores(pos_scores=empty, hard_neg_scores=empty, random_neg_scores=torch.randn(1, 4))}])
    for task in (Margin(), Softmax()):
        loss, n = task(ti, None, False, torch.device("cpu"))
        assert float(loss) == 0.0 and n == 0
