"""Count-min sketch: the oracle restatement against the reference's known-answer test, and the device sketch
(gigl_cms_add / gigl_cms_estimate) against the oracle — table cell for cell, estimates, and the Retrieval task's
candidate-sampling correction built on it."""
import numpy as np
import pytest
import torch

from oracle.cms import CountMinSketch as OracleCms


def test_oracle_reference_known_answers():
    # count_min_sketch_test.py:12-24
    cms = OracleCms(width=20, depth=5)
    cms.add_all(np.array([1, 2, 2, 3, 3, 3, 4, 4, 4, 4]))
    assert cms.total() == 10
    assert [cms.estimate(i) for i in (1, 2, 3, 4)] == [1, 2, 3, 4]
    assert cms.table.sum() == 10 * 5 and cms.table.dtype == np.int32


@pytest.mark.gpu
def test_device_sketch_reference_known_answers():
    from gigl_amd.count_min_sketch import CountMinSketch
    cms = CountMinSketch(width=20, depth=5)
    cms.add_torch_long_tensor(torch.tensor([1, 2, 2, 3, 3, 3, 4, 4, 4, 4], dtype=torch.long))
    assert cms.total() == 10
    assert [cms.estimate(i) for i in (1, 2, 3, 4)] == [1, 2, 3, 4]


@pytest.mark.gpu
@pytest.mark.parametrize("width,depth,n,hi", [(2000, 10, 5000, 300), (10000, 10, 20000, 1 << 40), (7, 3, 400, 50),
                                              (1, 1, 10, 5)])
def test_device_table_equals_the_oracle_table(width, depth, n, hi):
    """same cells as hash((id, row)) % width, including ids beyond 2^61 (the int hash wraps there), negative ids and
    -1 (whose int hash is -2)"""
    from gigl_amd.count_min_sketch import CountMinSketch
    rng = np.random.default_rng(width + depth)
    ids = rng.integers(0, hi, size=n, dtype=np.int64)
    ids[: min(8, n)] = np.array([0, -1, -2, (1 << 61) - 1, 1 << 61, (1 << 62) + 12345, -(1 << 62), 2 ** 63 - 1])[: min(8, n)]
    dev, ora = CountMinSketch(width=width, depth=depth), OracleCms(width=width, depth=depth)
    for part in np.array_split(ids, 3):  # several batches accumulate
        dev.add_torch_long_tensor(torch.from_numpy(part))
        ora.add_all(part)
    assert dev.total() == ora.total() == n
    np.testing.assert_array_equal(dev.get_table(), ora.table)
    probe = np.concatenate([ids[:500], rng.integers(0, max(hi, 2), size=200, dtype=np.int64)])
    got = dev.estimate_torch_long_tensor(torch.from_numpy(probe).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, ora.estimate_all(probe))
    true = {int(k): int(v) for k, v in zip(*np.unique(ids, return_counts=True))}
    assert all(g >= true.get(int(p), 0) for p, g in zip(probe, got))  # a count-min sketch never under-estimates


@pytest.mark.gpu
def test_retrieval_task_with_candidate_sampling_correction():
    """Retrieval(should_enable_candidate_sampling_correction=True) (task.py:140-205): logQ of the in-batch probability
    estimated by the two sketches is taken off the logits; equal to the loss evaluated with probabilities from the
    oracle sketch, and eval batches leave the sketches untouched"""
    from gigl_amd.count_min_sketch import calculate_in_batch_candidate_sampling_probability as in_batch_q
    from gigl_amd.link_prediction import RetrievalLoss
    from gigl_amd.nablp_spec import (BatchCombinedScores, BatchEmbeddings, NodeAnchorBasedLinkPredictionTaskInputs,
                                     Retrieval)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    task = Retrieval(temperature=0.1, should_enable_candidate_sampling_correction=True, count_min_sketch_width=50,
                     count_min_sketch_depth=4)
    ora_main, ora_rn = OracleCms(50, 4), OracleCms(50, 4)
    for step in range(3):
        nq, nh, nr = 12, 5, 9
        pos = torch.randint(0, 30, (nq,), generator=g)
        hard = torch.randint(0, 30, (nh,), generator=g)
        rn = torch.randint(0, 30, (nr,), generator=g)
        qid = torch.randint(100, 108, (nq,), generator=g)
        scores = torch.randn(nq, nq + nh + nr, generator=g)
        bcs = BatchCombinedScores(repeated_candidate_scores=scores.to(dev), positive_ids=pos.to(dev),
                                  hard_neg_ids=hard.to(dev), random_neg_ids=rn.to(dev), repeated_query_ids=qid.to(dev),
                                  num_unique_query_ids=int(qid.unique().numel()))
        emb = torch.zeros((nq, 4), device=dev)
        ti = NodeAnchorBasedLinkPredictionTaskInputs(
            main_batch=None, random_neg_batch=None,
            batch_embeddings=BatchEmbeddings(emb, {0: emb}, {0: emb}, {0: emb}, {0: emb}), batch_combined_scores={0: bcs})
        loss, n = task(ti, None, should_eval=False, device=dev)
        ora_main.add_all(pos.numpy())
        ora_main.add_all(hard.numpy())
        ora_rn.add_all(rn.numpy())
        prob = torch.cat((
            in_batch_q(torch.from_numpy(ora_main.estimate_all(pos.numpy())), ora_main.total(), nq + nh),
            in_batch_q(torch.from_numpy(ora_main.estimate_all(hard.numpy())), ora_main.total(), nq + nh),
            in_batch_q(torch.from_numpy(ora_rn.estimate_all(rn.numpy())), ora_rn.total(), nr)))
        want = RetrievalLoss(temperature=0.1, remove_accidental_hits=True).calculate_batch_retrieval_loss(
            scores.to(dev), candidate_sampling_probability=prob.to(dev), query_ids=qid.to(dev),
            candidate_ids=torch.cat((pos, hard, rn)).to(dev))
        plain = RetrievalLoss(temperature=0.1, remove_accidental_hits=True).calculate_batch_retrieval_loss(
            scores.to(dev), query_ids=qid.to(dev), candidate_ids=torch.cat((pos, hard, rn)).to(dev))
        assert n == nq and abs(float(loss) - float(want)) <= 1e-6 * abs(float(want))
        if step > 0:  # (first batch: every estimated probability is capped at 1, logQ = 0)
            assert abs(float(loss) - float(plain)) > 1e-3  # the correction does something
        before = task.main_batch_cm_sketch.total()
        ev, _ = task(ti, None, should_eval=True, device=dev)
        assert task.main_batch_cm_sketch.total() == before and abs(float(ev) - float(plain)) <= 1e-6 * abs(float(plain))
    np.testing.assert_array_equal(task.main_batch_cm_sketch.get_table(), ora_main.table)
