"""shared test helpers: fixture graph loading, validity predicates (the reference's own definition of a
correct sample — scala/subgraph_sampler/src/test/scala/SGSPureSparkV1TaskTest.scala:190-213,241-271,
436-505,509-580 and python/tests/integration/pipeline/subgraph_sampler/subgraph_sampler_test.py:529-684,
777-966), seeded synthetic graphs."""
import os

import numpy as np

from gigl_amd import wire

INVALID = 0xFFFFFFFF
A = "ref_assets"


def load_fixture_graph(golden_dir, task="supervised_node_classification"):
    """the 16-node / 34-directed-edge input tables of the reference's sampler tests"""
    nd = os.path.join(golden_dir, A, f"subgraph_sampler/{task}/node_data/data.tfrecord")
    ed = os.path.join(golden_dir, A, f"subgraph_sampler/{task}/edge_data/data.tfrecord")
    nodes = [wire.decode_tf_example(r) for r in wire.read_tfrecords(nd)]
    edges = [wire.decode_tf_example(r) for r in wire.read_tfrecords(ed)]
    n = len(nodes)
    feats = np.zeros((n, 2), dtype=np.float32)
    for r in nodes:
        feats[int(r["node_id"][0])] = [float(r["f0"][0]), float(r["f1"][0])]
    src = np.array([int(e["src"][0]) for e in edges], dtype=np.uint32)
    dst = np.array([int(e["dst"][0]) for e in edges], dtype=np.uint32)
    return n, src, dst, feats


def rmat_edges(scale, n_edges, seed, a=0.57, b=0.19, c=0.19):
    """seeded R-MAT edge list (numpy; small sizes for CPU tests)"""
    rng = np.random.default_rng(seed)
    src = np.zeros(n_edges, dtype=np.int64)
    dst = np.zeros(n_edges, dtype=np.int64)
    for _ in range(scale):
        r = rng.random(n_edges)
        q_b = (r >= a) & (r < a + b)
        q_c = (r >= a + b) & (r < a + b + c)
        q_d = r >= a + b + c
        src = src * 2 + (q_c | q_d)
        dst = dst * 2 + (q_b | q_d)
    return src.astype(np.uint32), dst.astype(np.uint32)


def check_rnn_validity(root, edges, nodes, rowptr, col, fanout, hops=2, exact_counts=True):
    """edges: iterable of (src, dst) global ids of one rooted neighbourhood"""
    edges = list(edges)
    nodes = list(nodes)
    nbrs = lambda v: set(int(x) for x in col[rowptr[v]:rowptr[v + 1]])
    assert root in nodes, "root must be in its neighbourhood"
    assert len(set(nodes)) == len(nodes), "duplicate nodes"
    assert len(set(edges)) == len(edges), "duplicate edges"
    ns = set(nodes)
    indeg = {}
    for s, d in edges:
        assert s in nbrs(d), f"edge {s}->{d} not in graph"
        assert s in ns and d in ns, "edge endpoint missing from nodes"
        indeg[d] = indeg.get(d, 0) + 1
    # in-degree per dst <= sum of fanouts over the hops at which it can appear, and <= true degree
    for d, c in indeg.items():
        assert c <= len(nbrs(d))
        assert c <= fanout * hops
    deg_r = len(nbrs(root))
    if deg_r == 0:
        assert not edges and nodes == [root]
        return
    hop1 = [s for s, d in edges if d == root]
    assert len(hop1) >= min(fanout, deg_r) or not exact_counts
    if exact_counts and deg_r <= fanout:
        # unambiguous case: every in-neighbour sampled, and each contributes min(f, deg) hop-2 edges
        assert set(hop1) == nbrs(root)
        want = set((a, root) for a in nbrs(root))
        for a in nbrs(root):
            if len(nbrs(a)) <= fanout:
                want |= set((b, a) for b in nbrs(a))
        assert want <= set(edges)


def adam_state_errors(got_params, got_moments, ref_params, ref_moments, determined=1e-3):
    """How far a trained state is from the reference's, in the quantities that CAN be compared after several Adam steps.
    Adam moves a parameter by lr * m_hat / (sqrt(v_hat) + eps): where a gradient element is rounding noise (|g| ~ 1e-9 of
    the tensor's largest) the quotient is +-1 whatever the magnitude, and two correct implementations drift apart by up
    to lr per step — raw parameters of such elements say nothing.  Compared instead, per tensor:
      * exp_avg / exp_avg_sq (linear / quadratic in the gradients), as max |diff| / max |reference| — every element;
      * the parameters on the DETERMINED set: elements whose gradient RMS sqrt(exp_avg_sq) is >= `determined` of the
        tensor's largest (the reference's moments decide the set), as max |diff|.
    *_params: name -> tensor; *_moments: name -> (exp_avg, exp_avg_sq).  -> {name: (err_m, err_v, err_param, share of the
    tensor that is determined)}"""
    import torch
    out = {}
    for k, (mr, vr) in ref_moments.items():
        m, v = got_moments[k]
        mr, vr, m, v = (t.detach().double().cpu().reshape(-1) for t in (mr, vr, m, v))
        p, pr = got_params[k].detach().double().cpu().reshape(-1), ref_params[k].detach().double().cpu().reshape(-1)
        rms = vr.sqrt()
        det = rms >= determined * rms.max()
        out[k] = (float((m - mr).abs().max() / (mr.abs().max() + 1e-300)), float((v - vr).abs().max() / (vr.max() + 1e-300)),
                  float((p - pr).abs()[det].max()) if bool(det.any()) else 0.0, float(det.double().mean()))
    return out


def torch_adam_moments(opt, named_params):
    """name -> (exp_avg, exp_avg_sq) of a torch.optim.Adam over `named_params` (name -> parameter)"""
    return {k: (opt.state[p]["exp_avg"], opt.state[p]["exp_avg_sq"]) for k, p in named_params.items() if p in opt.state}


def assert_adam_state(tag, got_params, got_moments, ref_params, ref_moments, tol_m, tol_v, tol_p):
    errs = adam_state_errors(got_params, got_moments, ref_params, ref_moments)
    worst = (max(e[0] for e in errs.values()), max(e[1] for e in errs.values()), max(e[2] for e in errs.values()),
             min(e[3] for e in errs.values()))
    print(f"{tag}: Adam state vs the reference: exp_avg {worst[0]:.2e} exp_avg_sq {worst[1]:.2e} of the tensors' largest, "
          f"parameters on the determined set {worst[2]:.2e} (smallest determined share {worst[3]:.2f})")
    for k, (em, ev, ep, share) in errs.items():
        assert em <= tol_m and ev <= tol_v and ep <= tol_p, (tag, k, em, ev, ep, share)
