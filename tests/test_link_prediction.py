"""Link-prediction head against the reference's known answers
(python/tests/unit/src/common/models/layers/loss_test.py:61-166, decoder_test.py:44-86)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gigl_amd.link_prediction import DecoderType, LinkPredictionDecoder, RetrievalLoss

Q = F.normalize(torch.tensor([[1.0, 1.0, 0.0, 0.0], [1.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.9, 1.1], [0.0, 0.0, 0.9, 1.1]]), p=2, dim=1)
POS = F.normalize(torch.tensor([[1.0, 0.2, 0.0, 0.0], [0.3, 1.0, 0.2, 0.0], [0.0, 0.0, 1.0, 0.4], [0.0, 0.0, 0.4, 0.9]]), p=2, dim=1)
NEG = F.normalize(torch.tensor([[0.21, 0.22, 0.23, 0.24], [0.24, 0.23, 0.22, 0.21]]), p=2, dim=1)
CAND = torch.concat([POS, NEG], dim=0)
CAND_IDS = torch.tensor([1, 2, 3, 4, 1, 5], dtype=torch.int64)  # one random negative collides with a positive
QUERY_IDS = torch.tensor([11, 11, 12, 12], dtype=torch.int64)   # each anchor has two positives
LABELS = torch.eye(4, 6)
MINF = torch.finfo(torch.float).min


def _masked(loss, scores, **kw):
    """the masked logits the fused kernel feeds to the cross-entropy (what the reference builds as tensors)"""
    got = {}

    class Capture(torch.nn.Module):
        def forward(self, inp, target):
            got["logits"], got["target"] = inp.detach().cpu(), target.detach().cpu()
            return inp.sum() * 0

    probe = RetrievalLoss(loss=Capture(), temperature=loss._temperature,
                          remove_accidental_hits=loss._remove_accidental_hits)
    probe.calculate_batch_retrieval_loss(scores, **kw)
    return got["logits"], got["target"]


@pytest.mark.gpu
def test_masks_match_the_reference_tests():
    """loss_test.py:61-113: the duplicate masks by query id and by candidate id, read off the masked logits"""
    dev = torch.device("cuda", 0)
    scores = torch.mm(Q, CAND.T).to(dev)
    logits, target = _masked(RetrievalLoss(), scores, query_ids=QUERY_IDS.to(dev))
    assert torch.equal(target, LABELS)
    want_q = torch.tensor([[1., 1, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0], [0, 0, 1, 1, 0, 0], [0, 0, 1, 1, 0, 0]])
    assert torch.equal((logits < -1e30).float(), want_q - LABELS)  # masked = duplicates minus the positive itself
    logits, _ = _masked(RetrievalLoss(remove_accidental_hits=True), scores, candidate_ids=CAND_IDS.to(dev))
    want_c = torch.tensor([[1., 0, 0, 0, 1, 0], [0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 0, 0]])
    assert torch.equal((logits < -1e30).float(), want_c - LABELS)
    assert torch.allclose(logits[logits > -1e30], scores.cpu()[logits > -1e30])


@pytest.mark.gpu
def test_loss_values():
    """loss_test.py:115-166 known answers, on the device"""
    dev = torch.device("cuda", 0)
    scores = torch.mm(Q, CAND.T).to(dev)
    e1 = torch.tensor([[0.8321, 0.8647, 0.0, 0.0, 0.6748, 0.7376], [0.8321, 0.8647, 0.0, 0.0, 0.6748, 0.7376],
                       [0.0, 0.1191, 0.8754, 0.9644, 0.7355, 0.6699], [0.0, 0.1191, 0.8754, 0.9644, 0.7355, 0.6699]])
    a1 = RetrievalLoss(remove_accidental_hits=False).calculate_batch_retrieval_loss(scores).cpu()
    assert torch.isclose(F.cross_entropy(e1, LABELS, reduction="sum"), a1, atol=1e-3)
    loss = RetrievalLoss(remove_accidental_hits=True)
    e2 = e1.clone()
    e2[0, 4] = MINF
    a2 = loss.calculate_batch_retrieval_loss(scores, candidate_ids=CAND_IDS.to(dev)).cpu()
    assert torch.isclose(F.cross_entropy(e2, LABELS, reduction="sum"), a2, atol=1e-3)
    e3 = e2.clone()
    e3[0, 1] = e3[1, 0] = e3[2, 3] = e3[3, 2] = MINF
    a3 = loss.calculate_batch_retrieval_loss(scores, candidate_ids=CAND_IDS.to(dev), query_ids=QUERY_IDS.to(dev)).cpu()
    assert torch.isclose(F.cross_entropy(e3, LABELS, reduction="sum"), a3, atol=1e-3)
    assert a3 < a2 < a1 + 1e-6  # masking other positives / accidental hits can only lower the loss
    with pytest.raises(ValueError):
        loss.calculate_batch_retrieval_loss(scores)  # accidental-hit removal needs candidate ids
    t = RetrievalLoss(temperature=0.07).calculate_batch_retrieval_loss(scores).cpu()
    assert torch.isclose(t, F.cross_entropy(scores.cpu() / 0.07, LABELS, reduction="sum"), rtol=1e-5)
    # sampled-softmax correction (loss.py:240-246) against the same algebra in torch
    prob = torch.tensor([0.5, 0.25, 1e-12, 0.125, 0.0625, 0.03125])
    c = RetrievalLoss().calculate_batch_retrieval_loss(scores, candidate_sampling_probability=prob.to(dev)).cpu()
    want = F.cross_entropy(scores.cpu() - torch.log(torch.clamp(prob, min=1e-10)), LABELS, reduction="sum")
    assert torch.isclose(c, want, rtol=1e-5)
    # the oracle's row-by-row restatement is pinned on the same known answers
    from oracle import gnn_ref
    sc = scores.cpu()
    o2 = gnn_ref.retrieval_loss_rows(sc, list(range(4)), CAND_IDS.tolist(), temperature=1.0)
    assert torch.isclose(F.cross_entropy(e2, LABELS, reduction="sum"), o2, atol=1e-3)
    o3 = gnn_ref.retrieval_loss_rows(sc, QUERY_IDS.tolist(), CAND_IDS.tolist(), temperature=1.0)
    assert torch.isclose(F.cross_entropy(e3, LABELS, reduction="sum"), o3, atol=1e-3) and torch.isclose(o3, a3)
    o1 = gnn_ref.retrieval_loss_rows(sc, list(range(4)), CAND_IDS.tolist(), temperature=1.0,
                                     remove_accidental_hits=False)
    assert torch.isclose(o1, a1)


@pytest.mark.gpu
def test_fused_loss_matches_the_oracle_and_autograd_at_size():
    """random [Q, C] scores with colliding ids: value vs the oracle's row-by-row restatement, gradient vs torch
    autograd through the same masked cross-entropy"""
    from oracle import gnn_ref
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    q, c = 37, 301
    scores = torch.randn(q, c, generator=g)
    qid = torch.randint(0, 12, (q,), generator=g)
    cid = torch.randint(0, 60, (c,), generator=g)
    loss = RetrievalLoss(temperature=0.07, remove_accidental_hits=True)
    s_dev = scores.to(dev).requires_grad_(True)
    got = loss.calculate_batch_retrieval_loss(s_dev, query_ids=qid.to(dev), candidate_ids=cid.to(dev))
    want = gnn_ref.retrieval_loss_rows(scores, qid.tolist(), cid.tolist(), temperature=0.07)
    assert abs(float(got) - float(want)) <= 1e-5 * abs(float(want))
    (got * 0.5).backward()
    s_ref = scores.clone().requires_grad_(True)
    (gnn_ref.retrieval_loss_rows(s_ref, qid.tolist(), cid.tolist(), temperature=0.07) * 0.5).backward()
    np.testing.assert_allclose(s_dev.grad.cpu().numpy(), s_ref.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_loss_construction_and_cpu_inputs():
    with pytest.raises(ValueError):
        RetrievalLoss(temperature=1e-13)
    with pytest.raises(RuntimeError):  # device tensors only: no CPU fallback
        RetrievalLoss().calculate_batch_retrieval_loss(torch.mm(Q, CAND.T))


def test_decoder_construction_errors():
    with pytest.raises(AttributeError):
        LinkPredictionDecoder(decoder_type="outer_product", decoder_channel_list=None)
    with pytest.raises(ValueError):
        LinkPredictionDecoder(decoder_type=DecoderType.hadamard_MLP, decoder_channel_list=None)
    with pytest.raises(ValueError):
        LinkPredictionDecoder(decoder_type=DecoderType.hadamard_MLP, decoder_channel_list=[1])
    with pytest.raises(ValueError):
        LinkPredictionDecoder(decoder_type=DecoderType.hadamard_MLP, decoder_channel_list=[2, 2])
    dec = LinkPredictionDecoder(decoder_type=DecoderType.inner_product, decoder_channel_list=[4, 2, 1])
    with pytest.raises(RuntimeError):  # no CPU fallback for the GEMM
        dec(Q, CAND)


@pytest.mark.gpu
def test_inner_product_decoder_known_answer_and_grads():
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    dev = eng.device
    q = F.normalize(torch.tensor([[1.0, 1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.9, 1.1], [0.0, 0.0, 0.9, 1.1],
                                  [0.0, 0.0, 0.1, 1.1]]), p=2, dim=1)
    c = torch.tensor([[0.9806, 0.1961, 0.0, 0.0], [0.2822, 0.9407, 0.1881, 0.0], [0.0, 0.0, 0.9285, 0.3714],
                      [0.0, 0.0, 0.4061, 0.9138], [0.4661, 0.4883, 0.5105, 0.5327], [0.5327, 0.5105, 0.4883, 0.4661]])
    want = torch.tensor([[0.8321, 0.8647, 0.0, 0.0, 0.6749, 0.7377], [0.9806, 0.2822, 0.0, 0.0, 0.4661, 0.5327],
                         [0.0, 0.1191, 0.8754, 0.9644, 0.7356, 0.6700], [0.0, 0.1191, 0.8754, 0.9644, 0.7356, 0.6700],
                         [0.0, 0.0170, 0.4539, 0.9468, 0.5767, 0.5084]])
    dec = LinkPredictionDecoder(decoder_type=DecoderType.inner_product, decoder_channel_list=[4, 2, 1])
    dec.engine = eng
    got = dec(q.to(dev), c.to(dev))
    assert got.shape.numel() == 5 * 6
    assert torch.allclose(got.cpu(), want, atol=1e-4)  # decoder_test.py:44-62
    # gradients == torch.mm autograd
    g = torch.Generator().manual_seed(0)
    qq, cc = torch.randn(37, 24, generator=g), torch.randn(53, 24, generator=g)
    q1, c1 = qq.clone().requires_grad_(True), cc.clone().requires_grad_(True)
    q2, c2 = qq.to(dev).requires_grad_(True), cc.to(dev).requires_grad_(True)
    wgt = torch.randn(37, 53, generator=g)
    (torch.mm(q1, c1.T) * wgt).sum().backward()
    (dec(q2, c2) * wgt.to(dev)).sum().backward()
    np.testing.assert_allclose(q2.grad.cpu().numpy(), q1.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c2.grad.cpu().numpy(), c1.grad.numpy(), rtol=1e-5, atol=1e-5)
    # end to end: decoder + retrieval loss on the device
    loss = RetrievalLoss(temperature=0.07, remove_accidental_hits=True)
    s = dec(Q.to(dev), CAND.to(dev))
    got_l = loss.calculate_batch_retrieval_loss(s, query_ids=QUERY_IDS.to(dev), candidate_ids=CAND_IDS.to(dev), device=dev)
    from oracle import gnn_ref
    ref_l = gnn_ref.retrieval_loss_rows(torch.mm(Q, CAND.T), QUERY_IDS.tolist(), CAND_IDS.tolist(), temperature=0.07)
    assert abs(float(got_l) - float(ref_l)) < 1e-4
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("channels,bias,plain_last", [([4, 2, 1], False, False), ([32, 16, 8, 1], True, False),
                                                      ([32, 16, 1], [True, False], True)])
def test_hadamard_mlp_decoder_scores_and_grads(channels, bias, plain_last):
    """scores[q, c] = MLP(q * c).sum(-1) (decoder.py:67-69; decoder_test.py:44-50 checks the [Q, C] shape) against the
    same MLP evaluated pair by pair with torch ops, forward and gradients"""
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    try:
        torch.manual_seed(len(channels))
        dec = LinkPredictionDecoder(decoder_type=DecoderType.hadamard_MLP, decoder_channel_list=channels, bias=bias,
                                    plain_last=plain_last).to(eng.device)
        dec.engine = eng
        d = channels[0]
        q = (Q if d == 4 else torch.randn(5, d) / 3).clone()
        c = (CAND if d == 4 else torch.randn(6, d) / 3).clone()
        qd, cd = q.to(eng.device).requires_grad_(True), c.to(eng.device).requires_grad_(True)
        got = dec(qd, cd)
        assert tuple(got.shape) == (q.shape[0], c.shape[0])
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in dec.state_dict().items()}
        qr, cr = q.clone().requires_grad_(True), c.clone().requires_grad_(True)
        x = qr.unsqueeze(1) * cr
        n = len(channels) - 1
        for i in range(n):
            x = x @ sd[f"mlp_decoder.lins.{i}.weight"].T
            if f"mlp_decoder.lins.{i}.bias" in sd:
                x = x + sd[f"mlp_decoder.lins.{i}.bias"]
            if not (plain_last and i == n - 1):
                x = torch.relu(x)
        want = x.sum(-1)
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-6)
        w = torch.randn(q.shape[0], c.shape[0])
        (got * w.to(eng.device)).sum().backward()
        (want * w).sum().backward()
        for name, prm in dec.named_parameters():
            np.testing.assert_allclose(prm.grad.cpu().numpy(), sd[name].grad.numpy(), rtol=1e-4, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(qd.grad.cpu().numpy(), qr.grad.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(cd.grad.cpu().numpy(), cr.grad.numpy(), rtol=1e-4, atol=1e-6)
    finally:
        eng.close()
