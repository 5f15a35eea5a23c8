"""SamplingOp-DAG sampler on the device (gigl_rows_dedup + gigl_expand_frontier per op) against the per-root CPU
restatement of GraphDBSampler (oracle/dag_sampler.py) on a DBLP-shaped heterogeneous graph (author / paper / venue as
in the reference's heterogeneous fixtures): identical edge and node sets per root, plus the contract's properties."""
import numpy as np
import pytest
import torch

from gigl_amd import wire
from oracle import dag_sampler
from gigl_amd.graphdb_sampler import (INCOMING, OUTGOING, EdgeType, HipGraphDBSampler, SamplingOp, SamplingOpDAG)

pytestmark = pytest.mark.gpu

A2P = EdgeType("author", "author_to_paper", "paper")
P2A = EdgeType("paper", "paper_to_author", "author")
P2V = EdgeType("paper", "published_in", "venue")
NODE_TYPES = {"author": 0, "paper": 1, "venue": 2}
CET = {A2P: 0, P2A: 1, P2V: 2}


@pytest.fixture(scope="module")
def world():
    rng = np.random.default_rng(0)
    n = {"author": 3000, "paper": 5000, "venue": 40}
    k = 20000
    a = (rng.zipf(1.6, k) % n["author"]).astype(np.uint32)
    p = rng.integers(0, n["paper"], k).astype(np.uint32)
    edges = {A2P: (a, p), P2A: (p, a),
             P2V: (np.arange(0, n["paper"], 2, dtype=np.uint32), rng.integers(0, n["venue"], n["paper"] // 2).astype(np.uint32))}
    feats = {t: rng.standard_normal((n[t], d)).astype(np.float32) for t, d in (("author", 3), ("paper", 5), ("venue", 2))}
    s = HipGraphDBSampler(NODE_TYPES, n, edges, CET, feats)
    yield s, n, edges, feats, dag_sampler.neighbour_lists(edges)
    s.close()


def _dag_two_paths():
    # root paper: its authors (op0) and its venue (op1); op2 = other papers of those authors; op3 (two parents, mixed
    # directions) = papers of the venue UNITED with op2's papers -> their authors
    return [SamplingOp("op0", A2P, 4, [], INCOMING),            # authors writing the root paper
            SamplingOp("op1", P2V, 1, [], OUTGOING),            # the root paper's venue
            SamplingOp("op2", A2P, 3, ["op0"], OUTGOING),       # papers of those authors
            SamplingOp("op3", P2V, 5, ["op1"], INCOMING),       # papers published in that venue
            SamplingOp("op4", P2A, 2, ["op2", "op3"], OUTGOING)]  # authors of the united paper set


def _check(world, ops, root_type, roots):
    s, n, edges, feats, nbrs = world
    dag = SamplingOpDAG.from_ops(ops)
    msgs = s.getKHopSubgraphForRootNodes(roots, root_type, dag)
    assert len(msgs) == len(roots)
    n_multi = 0
    for r, m in zip(roots, msgs):
        want_e, want_n = dag_sampler.sample_for_root(int(r), ops, nbrs, NODE_TYPES, CET, root_type)
        got_e = {(e.src_node_id, e.dst_node_id, e.condensed_edge_type) for e in m.neighborhood.edges}
        got_n = {(x.node_id, x.condensed_node_type) for x in m.neighborhood.nodes}
        assert got_e == want_e and got_n == want_n
        assert len(m.neighborhood.edges) == len(got_e) and len(m.neighborhood.nodes) == len(got_n)  # sets
        assert (m.root_node.node_id, m.root_node.condensed_node_type) == (int(r), NODE_TYPES[root_type])
        by_cnt = {c: t for t, c in NODE_TYPES.items()}
        for x in m.neighborhood.nodes:  # hydrated with the features of its own type
            np.testing.assert_array_equal(x.feature_values, feats[by_cnt[x.condensed_node_type]][x.node_id])
        n_multi += len(want_e) > 8
    return msgs, n_multi


def test_dag_matches_the_per_root_restatement(world):
    rng = np.random.default_rng(1)
    roots = rng.integers(0, 5000, 96)
    roots[:4] = [0, 2, 4998, 4999]
    msgs, n_multi = _check(world, _dag_two_paths(), "paper", roots)
    assert n_multi > 30


def test_chain_dag_and_skipped_paths(world):
    s, n, edges, feats, nbrs = world
    # an author root: papers (op0) -> venues (op1, OUTGOING) and co-authors (op2); odd papers have no venue: for such
    # roots op1 returns nothing and op3 (child of op1) must not run at all
    ops = [SamplingOp("op0", A2P, 2, [], OUTGOING), SamplingOp("op1", P2V, 1, ["op0"], OUTGOING),
           SamplingOp("op2", P2A, 3, ["op0"], OUTGOING), SamplingOp("op3", P2V, 4, ["op1", "op2"], INCOMING)]
    rng = np.random.default_rng(2)
    roots = rng.integers(0, 3000, 128)
    msgs, _ = _check(world, ops, "author", roots)
    dag = SamplingOpDAG.from_ops(ops)
    assert dag.execution_order() == ["op0", "op1", "op2", "op3"] and dag.root_op_names == ["op0"]
    # properties of the contract: at most n neighbours per frontier node and op; frontier rows are sets
    res = s.run_dag(torch.tensor(roots.astype(np.int32)), dag)
    for name, r in res.items():
        f = dag.nodes[name].sampling_op.num_nodes_to_sample
        cnt = r.cnt.cpu().numpy()
        assert cnt.max() <= f
        fr = r.frontier.cpu().numpy().view(np.uint32)
        for row in fr:
            v = row[row != 0xFFFFFFFF]
            assert len(np.unique(v)) == len(v)
    # some roots have no venue on any of their papers: op3 contributed nothing for them although op2 returned nodes
    lonely = [m for m in msgs if not any(e.condensed_edge_type == CET[P2V] for e in m.neighborhood.edges)]
    assert lonely and all(all(x.condensed_node_type != NODE_TYPES["venue"] for x in m.neighborhood.nodes) for m in lonely)


def test_rows_dedup_kernel(world):
    s = world[0]
    eng = s.engine
    rng = np.random.default_rng(3)
    for rows, width in ((1, 1), (7, 33), (300, 700), (4, 8192)):
        a = rng.integers(0, max(width // 3, 2), (rows, width)).astype(np.uint32)
        a[rng.random((rows, width)) < 0.2] = 0xFFFFFFFF
        t = torch.from_numpy(a.view(np.int32)).to(eng.device).contiguous()
        eng.rows_dedup(t)
        got = t.cpu().numpy().view(np.uint32)
        for r in range(rows):
            want = a[r].copy()
            seen = set()
            for q, v in enumerate(a[r].tolist()):
                if v == 0xFFFFFFFF:
                    continue
                if v in seen:
                    want[q] = 0xFFFFFFFF
                seen.add(v)
            np.testing.assert_array_equal(got[r], want)
    from gigl_amd import _lib
    with pytest.raises(_lib.GiglError):
        eng.rows_dedup(torch.zeros((2, 9000), dtype=torch.int32, device=eng.device))


def test_reference_heterogeneous_fixture(golden_dir):
    """the reference's own heterogeneous sampler fixture (scala/common/src/test/assets/subgraph_sampler/heterogeneous/
    node_anchor_based_link_prediction: users / stories named author / paper in its graph metadata, edge types
    author_to_paper = 0 and paper_to_author = 1, tf.Example tables with f0, f1) read by the native ingest and sampled
    with a two-hop op DAG per root type: identical to the per-root restatement, hydrated with the typed features"""
    import os
    from gigl_amd.ingest import COL_F32, COL_I64, read_columns
    base = os.path.join(golden_dir, "ref_assets/subgraph_sampler/heterogeneous/node_anchor_based_link_prediction")

    def table(rel, cols):
        data, _ = read_columns([os.path.join(base, rel, "data.tfrecord")], cols)
        return data
    nodes = {"author": table("node_features_dir/user/features", [("node_id", COL_I64, 1), ("f0", COL_F32, 1), ("f1", COL_F32, 1)]),
             "paper": table("node_features_dir/story/features", [("node_id", COL_I64, 1), ("f0", COL_F32, 1), ("f1", COL_F32, 1)])}
    a2p, p2a = EdgeType("author", "author_to_paper", "paper"), EdgeType("paper", "paper_to_author", "author")
    ecols = [("src", COL_I64, 1), ("dst", COL_I64, 1), ("f0", COL_F32, 1), ("f1", COL_F32, 1)]
    et_tables = {a2p: table("edge_features_dir/user-to-story/main_edges/features", ecols),
                 p2a: table("edge_features_dir/story-to-user/main_edges/features", ecols)}
    n = {t: int(d["node_id"].max()) + 1 for t, d in nodes.items()}
    assert n == {"author": 15, "paper": 19}
    feats = {}
    for t, d in nodes.items():
        x = np.zeros((n[t], 2), np.float32)
        x[d["node_id"][:, 0]] = np.concatenate([d["f0"], d["f1"]], axis=1)
        feats[t] = x
    edges = {et: (d["src"][:, 0].astype(np.uint32), d["dst"][:, 0].astype(np.uint32)) for et, d in et_tables.items()}
    for et, (s_, d_) in edges.items():  # ids stay inside their own type's id space
        assert s_.max() < n[et.src_node_type] and d_.max() < n[et.dst_node_type]
    types, cet = {"author": 0, "paper": 1}, {a2p: 0, p2a: 1}
    efeats = {et: np.concatenate([d["f0"], d["f1"]], axis=1) for et, d in et_tables.items()}
    s = HipGraphDBSampler(types, n, edges, cet, feats, edge_features=efeats)
    nbrs = dag_sampler.neighbour_lists(edges)
    plans = {"paper": [SamplingOp("h1", a2p, 3, [], INCOMING), SamplingOp("h2", p2a, 3, ["h1"], INCOMING)],
             "author": [SamplingOp("h1", p2a, 3, [], INCOMING), SamplingOp("h2", a2p, 3, ["h1"], INCOMING),
                        SamplingOp("out", a2p, 2, [], OUTGOING)]}
    total_edges = 0
    for root_type, ops in plans.items():
        roots = np.arange(n[root_type])
        msgs = s.getKHopSubgraphForRootNodes(roots, root_type, SamplingOpDAG.from_ops(ops))
        by_cnt = {c: t for t, c in types.items()}
        for r, m in zip(roots, msgs):
            want_e, want_n = dag_sampler.sample_for_root(int(r), ops, nbrs, types, cet, root_type)
            assert {(e.src_node_id, e.dst_node_id, e.condensed_edge_type) for e in m.neighborhood.edges} == want_e
            assert {(x.node_id, x.condensed_node_type) for x in m.neighborhood.nodes} == want_n
            for x in m.neighborhood.nodes:
                np.testing.assert_array_equal(x.feature_values, feats[by_cnt[x.condensed_node_type]][x.node_id])
            for e in m.neighborhood.edges:  # every sampled edge is an edge of its type's table
                et = a2p if e.condensed_edge_type == 0 else p2a
                src, dst = edges[et]
                hit = np.flatnonzero((src == e.src_node_id) & (dst == e.dst_node_id))
                assert hit.size > 0
                np.testing.assert_array_equal(e.feature_values, efeats[et][hit[0]])  # the edge table's own row
            total_edges += len(want_e)
        # the same messages written by the device encoder, edge features included: byte for byte
        dev_recs = s.encode_records(roots, root_type, SamplingOpDAG.from_ops(ops), tfrecord_frame=False)
        assert [m.SerializeToString() for m in msgs] == dev_recs
        assert any(e.feature_values.size for m in msgs for e in m.neighborhood.edges)
    assert total_edges > 100
    # typed TRAINING samples (GraphDBNodeAnchorBasedLinkPredictionTask.scala:333-470): supervision edge type
    # paper -> author; pos_edges sampled OUTGOING from the root, neighbourhood = the root's merged with its positives'
    root_dag = SamplingOpDAG.from_ops(plans["paper"])
    pos_ops = [op for op in plans["author"] if op.op_name != "out"]
    pos_dag = SamplingOpDAG.from_ops(pos_ops)
    with pytest.raises(ValueError):  # a root op of the positives' DAG must end in the positive node type
        s.getNablpSamplesForRootNodes([0], p2a, 1, root_dag, SamplingOpDAG.from_ops(plans["author"]))
    papers = np.arange(n["paper"])
    src_p2a, dst_p2a = edges[p2a]
    for P in (1, 3):
        samples = s.getNablpSamplesForRootNodes(papers, p2a, P, root_dag, pos_dag)
        recs, n_pos = s.encode_nablp_records(papers, p2a, P, root_dag, pos_dag, tfrecord_frame=False)
        assert [m.SerializeToString() for m in samples] == recs
        rnn_paper = s.getKHopSubgraphForRootNodes(papers, "paper", root_dag)
        seen_pos = 0
        for r, m, own, k in zip(papers, samples, rnn_paper, n_pos):
            out_nbrs = set(dst_p2a[src_p2a == r].tolist())
            assert len(m.pos_edges) == k == min(P, len(out_nbrs))
            got_n = {(x.node_id, x.condensed_node_type) for x in m.neighborhood.nodes}
            got_e = {(e.src_node_id, e.dst_node_id, e.condensed_edge_type) for e in m.neighborhood.edges}
            assert {(x.node_id, x.condensed_node_type) for x in own.neighborhood.nodes} <= got_n
            assert {(e.src_node_id, e.dst_node_id, e.condensed_edge_type) for e in own.neighborhood.edges} <= got_e
            want_n = {(x.node_id, x.condensed_node_type) for x in own.neighborhood.nodes}
            want_e = {(e.src_node_id, e.dst_node_id, e.condensed_edge_type) for e in own.neighborhood.edges}
            for e in m.pos_edges:
                assert e.src_node_id == r and e.dst_node_id in out_nbrs and e.condensed_edge_type == cet[p2a]
                hit = np.flatnonzero((src_p2a == r) & (dst_p2a == e.dst_node_id))
                np.testing.assert_array_equal(e.feature_values, efeats[p2a][hit[0]])
                pos_rnn = s.getKHopSubgraphForRootNode(e.dst_node_id, "author", pos_dag)  # the positive's own RNN
                want_n |= {(x.node_id, x.condensed_node_type) for x in pos_rnn.neighborhood.nodes}
                want_e |= {(x.src_node_id, x.dst_node_id, x.condensed_edge_type) for x in pos_rnn.neighborhood.edges}
                seen_pos += 1
            assert (got_n, got_e) == (want_n, want_e)  # mergeGraphs(root's, positives'), nothing else
            parsed = wire.NodeAnchorBasedLinkPredictionSample.FromString(recs[int(r)])
            assert parsed.root_node.node_id == r and len(parsed.pos_edges) == k
        assert seen_pos > 10
    s.close()


def test_typed_records_encoded_on_the_device_match_the_host_assembly(world):
    """gigl_typed_records_encode: the DAG's typed RootedNodeNeighborhood records written on the device are, byte for
    byte (TFRecord frame and CRCs included), the host assembly's messages; read back by the typed native collate they
    give the batch graph of HipGraphDBSampler.batch_graph."""
    from gigl_amd import wire
    from gigl_amd._lib import REC_ROOTED_NODE_NEIGHBORHOOD
    from gigl_amd.batches import collate_serialized_typed
    s, n, edges, feats, nbrs = world
    rng = np.random.default_rng(11)
    for ops, root_type, roots in ((_dag_two_paths(), "paper", rng.integers(0, 5000, 70)),
                                  ([SamplingOp("op0", A2P, 2, [], OUTGOING), SamplingOp("op1", P2V, 1, ["op0"], OUTGOING),
                                    SamplingOp("op2", P2A, 3, ["op0"], OUTGOING)], "author",
                                   np.concatenate([[0, 1, 2999], rng.integers(0, 3000, 61)]))):
        dag = SamplingOpDAG.from_ops(ops)
        msgs = s.getKHopSubgraphForRootNodes(roots, root_type, dag)
        framed = s.encode_records(roots, root_type, dag, tfrecord_frame=True)
        bare = s.encode_records(roots, root_type, dag, tfrecord_frame=False)
        assert len(framed) == len(msgs) == len(bare)
        for m, fr, br in zip(msgs, framed, bare):
            want = m.SerializeToString()
            assert br == want
            assert fr == wire.tfrecord_frame(want)
        # device records -> typed collate == the batch graph built on the device from the same samples
        ends = [None] * len(CET)
        for et, c in CET.items():
            ends[c] = (NODE_TYPES[et.src_node_type], NODE_TYPES[et.dst_node_type])
        col = collate_serialized_typed(bare, REC_ROOTED_NODE_NEIGHBORHOOD, len(NODE_TYPES), ends)
        g, root_index, uniq = s.batch_graph(roots, root_type, dag)
        for tname, c in NODE_TYPES.items():
            if tname not in uniq:
                assert col["node_ids"][c].size == 0
                continue
            ids = col["node_ids"][c].astype(np.int64)
            assert np.array_equal(np.sort(ids), uniq[tname].cpu().numpy())
            order = np.argsort(ids)  # collate numbers first-seen, the device graph ascending
            np.testing.assert_array_equal(col["x"][c][order], g.x_dict[tname].cpu().numpy())
        for et, c in CET.items():
            key = (et.src_node_type, et.relation, et.dst_node_type)
            ei = col["edge_index"][c]
            got = set(zip(col["node_ids"][NODE_TYPES[et.src_node_type]][ei[0]].tolist(),
                          col["node_ids"][NODE_TYPES[et.dst_node_type]][ei[1]].tolist())) if ei.size else set()
            if key not in g.edge_index_dict:
                assert not got
                continue
            e2 = g.edge_index_dict[key].cpu().numpy()
            want = set(zip(uniq[et.src_node_type].cpu().numpy()[e2[0]].tolist(),
                           uniq[et.dst_node_type].cpu().numpy()[e2[1]].tolist()))
            assert got == want
        assert np.array_equal(col["root_type"], np.full(len(roots), NODE_TYPES[root_type]))
        assert np.array_equal(col["node_ids"][NODE_TYPES[root_type]][col["root_local"]].astype(np.int64), np.asarray(roots))


def test_typed_records_beyond_the_lds_sort(world):
    """more than 4,095 sampled slots per root: the encoder's sort is staged in global scratch (typed_plan_kernel<true>,
    chunks of 4,096 through LDS) — records stay byte-identical to the host assembly: 4,160 slots per root on the skewed
    fixture (mostly duplicates), 16,448 on a dense one (> 8,000 distinct edges per record); the limit is 2^20 - 1"""
    from gigl_amd import _lib, wire
    rng = np.random.default_rng(3)
    nd = {"author": 5000, "paper": 5000, "venue": 4}
    a = rng.integers(0, 5000, 320000).astype(np.uint32)
    p_ = rng.integers(0, 5000, 320000).astype(np.uint32)
    dense_edges = {A2P: (a, p_), P2A: (p_, a), P2V: (np.arange(8, dtype=np.uint32), np.arange(8, dtype=np.uint32) % 4)}
    dense = HipGraphDBSampler(NODE_TYPES, nd, dense_edges, CET,
                              {t: rng.standard_normal((nd[t], d)).astype(np.float32) for t, d in (("author", 3), ("paper", 5), ("venue", 2))})
    cases = ((world[0], [SamplingOp("op0", A2P, 64, [], INCOMING), SamplingOp("op1", A2P, 64, ["op0"], OUTGOING)], "paper",
              [1, 2, 77, 4999], 100),
             (dense, [SamplingOp("op0", A2P, 64, [], OUTGOING), SamplingOp("op1", P2A, 64, ["op0"], OUTGOING),
                      SamplingOp("op2", A2P, 3, ["op1"], OUTGOING)], "author", [0, 1, 2500, 4999], 8000))
    try:
        for s, ops, root_type, roots, floor in cases:
            dag = SamplingOpDAG.from_ops(ops)
            msgs = s.getKHopSubgraphForRootNodes(roots, root_type, dag)
            framed = s.encode_records(roots, root_type, dag, tfrecord_frame=True)
            bare = s.encode_records(roots, root_type, dag, tfrecord_frame=False)
            assert max(len(m.neighborhood.edges) for m in msgs) > floor
            for m, fr, br in zip(msgs, framed, bare):
                want = m.SerializeToString()
                assert br == want
                assert fr == wire.tfrecord_frame(want)
    finally:
        dense.close()
    huge = [SamplingOp("op0", A2P, 64, [], INCOMING), SamplingOp("op1", A2P, 64, ["op0"], OUTGOING),
            SamplingOp("op2", P2A, 64, ["op1"], OUTGOING), SamplingOp("op3", A2P, 4, ["op2"], OUTGOING)]  # > 2^20 slots
    with pytest.raises(_lib.GiglError):
        world[0].encode_records([1, 2], "paper", SamplingOpDAG.from_ops(huge))


def test_one_call_typed_plan_equals_the_staged_batch_graph(world):
    """gigl_typed_plan_* (the DAG's ops, the per-type distinct ids and the per-edge-type distinct edges as one stream of
    device work, one host read of the counts) == batch_graph (one library call per op + torch.unique / searchsorted
    chains): same node numbering, same edge lists, same rows, same root positions — multi-parent ops, both directions,
    roots whose paths die out, repeated runs of one plan with different batch sizes"""
    s, n, edges, feats, nbrs = world
    chain = [SamplingOp("op0", A2P, 2, [], OUTGOING), SamplingOp("op1", P2V, 1, ["op0"], OUTGOING),
             SamplingOp("op2", P2A, 3, ["op0"], OUTGOING), SamplingOp("op3", P2V, 4, ["op1", "op2"], INCOMING)]
    for root_type, ops in (("paper", _dag_two_paths()), ("author", chain)):
        dag = SamplingOpDAG.from_ops(ops)
        rng = np.random.default_rng(7)
        for b in (1, 37, 256, 64):
            roots = rng.choice(n[root_type], size=b, replace=False)
            g0, ri0, u0 = s.batch_graph(roots, root_type, dag)
            g1, ri1, u1 = s.batch_graph_plan(roots, root_type, dag, b_max=256)
            torch.cuda.synchronize()
            assert set(u0) == set(u1), (root_type, b)
            for t in u0:
                assert torch.equal(u0[t], u1[t]), f"{root_type} b={b}: node list of {t}"
                assert torch.equal(g0.x_dict[t], g1.x_dict[t])
            assert set(g0.edge_index_dict) == set(g1.edge_index_dict)
            for k in g0.edge_index_dict:
                assert torch.equal(g0.edge_index_dict[k], g1.edge_index_dict[k]), f"{root_type} b={b}: edges of {k}"
            assert torch.equal(ri0, ri1)
        assert sum(int(v.shape[1]) for v in g1.edge_index_dict.values()) > 100


def test_ops_with_fanouts_beyond_64(world):
    """num_nodes_to_sample is any int in the reference (GraphDBSampler.scala:45-113): ops past the wave-resident
    selection's 64 — per-root sets equal to the restatement, the one-call plan equal to the staged path"""
    s, n, edges, feats, nbrs = world
    ops = [SamplingOp("op0", A2P, 100, [], OUTGOING), SamplingOp("op1", P2A, 2, ["op0"], OUTGOING),
           SamplingOp("op2", A2P, 70, ["op1"], OUTGOING)]
    deg = np.bincount(edges[A2P][0], minlength=n["author"])
    assert (deg > 100).sum() >= 4  # rows the selection really has to cut
    roots = np.concatenate([np.argsort(-deg)[:6], np.random.default_rng(3).integers(0, n["author"], 26)])
    roots = np.unique(roots)
    _check(world, ops, "author", roots)
    dag = SamplingOpDAG.from_ops(ops)
    g0, ri0, u0 = s.batch_graph(roots, "author", dag)
    g1, ri1, u1 = s.batch_graph_plan(roots, "author", dag, b_max=64)
    torch.cuda.synchronize()
    for t in u0:
        assert torch.equal(u0[t], u1[t]) and torch.equal(g0.x_dict[t], g1.x_dict[t])
    for k in g0.edge_index_dict:
        assert torch.equal(g0.edge_index_dict[k], g1.edge_index_dict[k])
    assert torch.equal(ri0, ri1)
    # and the device encoder's records of the same DAG
    bare = s.encode_records(roots, "author", dag, tfrecord_frame=False)
    msgs = s.getKHopSubgraphForRootNodes(roots, "author", dag)
    assert len(bare) == len(msgs) and all(b == m.SerializeToString() for b, m in zip(bare, msgs))


def test_one_call_typed_plan_at_scale():
    """the typed plan against the staged path on a 600k-node DBLP-shaped graph with skewed authors (hub rows in the
    heavy-row path of the sampler), 2,048 roots per batch: identical node lists, edge lists and root positions; repeated
    runs are identical; every root is in its type's list at its reported position"""
    rng = np.random.default_rng(0)
    na, npp, ne = 200_000, 400_000, 4_000_000
    a2p, p2a = EdgeType("author", "writes", "paper"), EdgeType("paper", "written_by", "author")
    src = (na * rng.random(ne) ** 2).astype(np.uint32)
    dst = rng.integers(0, npp, ne).astype(np.uint32)
    feats = {"author": rng.standard_normal((na, 8)).astype(np.float32), "paper": rng.standard_normal((npp, 12)).astype(np.float32)}
    s = HipGraphDBSampler({"author": 0, "paper": 1}, {"author": na, "paper": npp}, {a2p: (src, dst), p2a: (dst, src)},
                          {a2p: 0, p2a: 1}, feats)
    dag = SamplingOpDAG.from_ops([SamplingOp("h1", a2p, 10, [], INCOMING), SamplingOp("h2", p2a, 5, ["h1"], INCOMING)])
    roots = rng.choice(npp, size=2048, replace=False)
    g0, ri0, u0 = s.batch_graph(roots, "paper", dag)
    g1, ri1, u1 = s.batch_graph_plan(roots, "paper", dag)
    g2, ri2, u2 = s.batch_graph_plan(roots, "paper", dag)
    torch.cuda.synchronize()
    for t in u0:
        assert torch.equal(u0[t], u1[t]) and torch.equal(u1[t], u2[t])
        assert bool((u1[t][1:] > u1[t][:-1]).all())  # ascending, duplicate-free
        assert torch.equal(g0.x_dict[t], g1.x_dict[t])
    for k in g0.edge_index_dict:
        assert torch.equal(g0.edge_index_dict[k], g1.edge_index_dict[k]) and torch.equal(g1.edge_index_dict[k], g2.edge_index_dict[k])
    assert torch.equal(ri0, ri1) and torch.equal(u1["paper"][ri1].cpu(), torch.from_numpy(roots))
    assert sum(int(v.shape[1]) for v in g1.edge_index_dict.values()) > 50_000
    s.close()
