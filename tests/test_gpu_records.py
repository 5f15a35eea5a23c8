"""GPU parity of the device-side record encoder (gigl_records_encode) against oracle/records.py — the CPU restatement
of the sampler job's output stage (per-root assembly, proto3 encoding, TFRecord framing), which
tests/test_oracle_records.py pins byte for byte on the reference's own sampler output files.  (gigl_amd/wire.py appears
here only as a PARSER of the device's bytes; the expected bytes never come from product code.)

Bar: bit-exact (bytes of every TFRecord frame, including both masked CRC-32C words)."""
import os

import numpy as np
import pytest
import torch

import oracle
from gigl_amd import _lib, wire
from helpers import load_fixture_graph, rmat_edges
from oracle import records as R

pytestmark = pytest.mark.gpu


def _engine(n, src, dst, feats, directed=False):
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    eng.build_from_coo(n, src, dst, is_directed=directed)
    eng.build_from_coo(n, dst, src, is_directed=directed, out_graph=True)
    if feats is not None:
        eng.load_features(feats)
    return eng


def _host_rnn_frames(roots, fanouts, nbr, feats, node_type=0, edge_type=0, suffixes=None, emit=None):
    frames = []
    for i, (r, (s, d)) in enumerate(zip(roots.tolist(), R.tree_edges(roots, fanouts, nbr))):
        if emit is not None and not emit[i]:
            continue
        frames.append(R.tfrecord_frame(R.rooted_node_neighborhood_record(
            r, s, d, feats, node_type, edge_type, suffix=suffixes[i] if suffixes is not None else b"")))
    return frames


def _split(buf, off):
    off = off.cpu().numpy()
    b = buf.cpu().numpy().tobytes()
    return [b[off[i]:off[i + 1]] for i in range(len(off) - 1)]


@pytest.mark.parametrize("dtype", ["f32", "f16", "none"])
def test_rnn_records_match_host_codec_on_reference_fixture(golden_dir, dtype):
    n, src, dst, feats = load_fixture_graph(golden_dir)
    x = None if dtype == "none" else (feats if dtype == "f32" else torch.from_numpy(feats).to(torch.float16))
    eng = _engine(n, src, dst, x)
    roots = np.arange(n, dtype=np.uint32)  # includes node id 0 (elided field) and the two isolated nodes
    fanouts = [3, 3]
    tree = eng.sample_khop(roots, fanouts)
    buf, off = eng.encode_records(tree)
    nbr = [t.cpu().numpy().view(np.uint32) for t in tree.nbr]
    hx = None if dtype == "none" else (feats if dtype == "f32" else feats.astype(np.float16).astype(np.float32))
    want = _host_rnn_frames(roots, fanouts, nbr, hx)
    got = _split(buf, off)
    assert got == want
    # and the stream as a whole is a readable TFRecord file of valid messages
    recs = list(wire.iter_tfrecords(b"".join(got)))
    assert len(recs) == n
    for r, rec in zip(roots.tolist(), recs):
        m = wire.RootedNodeNeighborhood.FromString(rec)
        assert m.root_node.node_id == r
    eng.close()


@pytest.mark.parametrize("fanouts,d,node_type,edge_type", [([25, 10], 100, 0, 0), ([10, 5], 7, None, None),
                                                           ([4], 33, 3, 300), ([3, 2, 2], 1, 0, None)])
def test_rnn_records_random_graph(fanouts, d, node_type, edge_type):
    rng = np.random.default_rng(5)
    n = 300_000  # ids need 1..3 varint bytes
    src, dst = rmat_edges(18, 600_000, seed=11)
    src, dst = (src.astype(np.int64) * 2654435761 % n).astype(np.uint32), (dst.astype(np.int64) * 40503 % n).astype(np.uint32)
    feats = rng.standard_normal((n, d)).astype(np.float32)
    eng = _engine(n, src, dst, feats)
    roots = rng.integers(0, n, 96).astype(np.uint32)
    roots[:3] = [0, 1, n - 1]
    tree = eng.sample_khop(roots, fanouts)
    buf, off = eng.encode_records(tree, condensed_node_type=node_type, condensed_edge_type=edge_type)
    nbr = [t.cpu().numpy().view(np.uint32) for t in tree.nbr]
    want = _host_rnn_frames(roots, fanouts, nbr, feats, node_type, edge_type)
    got = _split(buf, off)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, f"record {i} (root {roots[i]}) differs: {len(g)} vs {len(w)} bytes"
    assert len(got) == len(want)
    # raw payloads (no framing)
    buf2, off2 = eng.encode_records(tree, condensed_node_type=node_type, condensed_edge_type=edge_type,
                                    tfrecord_frame=False)
    assert _split(buf2, off2) == [w[12:-4] for w in want]
    eng.close()


def test_supervised_samples_with_label_suffix_and_emit_mask(golden_dir):
    n, src, dst, feats = load_fixture_graph(golden_dir)
    eng = _engine(n, src, dst, feats)
    roots = np.arange(n, dtype=np.uint32)
    fanouts = [3, 3]
    tree = eng.sample_khop(roots, fanouts)
    labels = [R.encode_label("node_label", int(i % 5) - 1) for i in range(n)]  # incl. 0 and negative
    sfx = [b"" if i % 4 == 3 else R._ld(3, lb) for i, lb in enumerate(labels)]
    emit = np.array([1 if (i % 4 != 3 and i not in (14, 15)) else 0 for i in range(n)], dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(s) for s in sfx], out=off[1:])
    buf, rec_off = eng.encode_records(tree, emit=torch.from_numpy(emit),
                                      suffix=torch.from_numpy(np.frombuffer(b"".join(sfx), dtype=np.uint8).copy()),
                                      suffix_off=torch.from_numpy(off))
    nbr = [t.cpu().numpy().view(np.uint32) for t in tree.nbr]
    want = _host_rnn_frames(roots, fanouts, nbr, feats, suffixes=sfx, emit=emit)
    got = [g for g in _split(buf, rec_off) if g]
    assert got == want
    for rec in wire.iter_tfrecords(b"".join(got)):
        m = wire.SupervisedNodeClassificationSample.FromString(rec)
        assert len(m.root_node_labels) == 1 and m.root_node_labels[0].label_type == "node_label"
    eng.close()


def test_link_prediction_samples_match_host_assembly():
    """NodeAnchorBasedLinkPredictionSample: merged neighbourhoods of the root and of its positives
    (gigl_amd.subgraph_sampler.SubgraphSampler._run_nablp restated per record)"""
    rng = np.random.default_rng(9)
    n = 5000
    src, dst = rmat_edges(13, 40_000, seed=3)
    src, dst = (src % n).astype(np.uint32), (dst % n).astype(np.uint32)
    feats = rng.standard_normal((n, 5)).astype(np.float32)
    eng = _engine(n, src, dst, feats)
    roots = rng.integers(0, n, 64).astype(np.uint32)
    fanouts, P = [5, 3], 3
    pos, cnt = eng.sample_positives(roots, P)
    pos_h = pos.cpu().numpy().view(np.uint32).reshape(-1, P)
    cnt_h = cnt.cpu().numpy()
    all_roots = np.concatenate([roots[:, None], pos_h], axis=1).reshape(-1)  # root, its positives (INVALID padded)
    tree = eng.sample_khop(all_roots, fanouts)
    buf, off = eng.encode_records(tree, kind=_lib.REC_NODE_ANCHOR_LINK_PRED, trees_per_record=1 + P,
                                  emit=(cnt > 0).to(torch.uint8))
    nbr = [t.cpu().numpy().view(np.uint32) for t in tree.nbr]
    lists = R.tree_edges(all_roots, fanouts, nbr)
    want = []
    for i, r in enumerate(roots.tolist()):
        if cnt_h[i] == 0:
            continue
        c = int(cnt_h[i])
        want.append(R.tfrecord_frame(R.nablp_sample_record(
            r, [lists[i * (1 + P) + k] for k in range(1 + c)], pos_h[i, :c].tolist(), c, feats)))
    got = [g for g in _split(buf, off) if g]
    assert len(got) == len(want) and len(want) > 10
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, f"sample {i} differs"
    eng.close()


def _pair_features(de):
    """deterministic feature row of an undirected edge {a, b}: both directions and every duplicate input row agree"""
    def f(s, d):
        a, b = (s, d) if s < d else (d, s)
        return np.sin(np.arange(1, de + 1, dtype=np.float64) * (a * 0.37 + b * 1.13 + 1.0)).astype(np.float32)
    return f


@pytest.mark.parametrize("de,d,fanouts", [(1, 4, [5, 3]), (3, 0, [6, 2]), (40, 9, [25, 10]), (16, 3, [3, 2, 2])])
def test_rnn_records_with_edge_features(de, d, fanouts):
    """Edge.feature_values (hydrateEdges): every neighbourhood edge carries the row of its (src, dst) pair; de = 40
    pushes the Edge body past 127 bytes (two-byte length varint)"""
    rng = np.random.default_rng(de)
    n = 20_000
    src, dst = rmat_edges(15, 90_000, seed=21)
    src, dst = (src % n).astype(np.uint32), (dst.astype(np.int64) * 7 % n).astype(np.uint32)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    feats = rng.standard_normal((n, d)).astype(np.float32) if d else None
    f = _pair_features(de)
    eng = _engine(n, src, dst, feats)
    eng.load_edge_features(src, dst, np.stack([f(int(a), int(b)) for a, b in zip(src, dst)]), is_directed=False)
    roots = rng.integers(0, n, 80).astype(np.uint32)
    roots[:2] = [0, n - 1]
    tree = eng.sample_khop(roots, fanouts)
    buf, off = eng.encode_records(tree, with_features=d > 0)
    nbr = [t.cpu().numpy().view(np.uint32) for t in tree.nbr]
    want = []
    for r, (s_, d_) in zip(roots.tolist(), R.tree_edges(roots, fanouts, nbr)):
        want.append(R.tfrecord_frame(R.rooted_node_neighborhood_record(r, s_, d_, feats, edge_features=f)))
    got = _split(buf, off)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, f"record {i} differs"
    big = max(got, key=len)
    m = wire.RootedNodeNeighborhood.FromString(next(iter(wire.iter_tfrecords(big))))
    assert all(e.feature_values.size == de for e in m.neighborhood.edges) and len(m.neighborhood.edges) > 0
    # without the table attached the same call writes feature-less edges
    plain, _ = eng.encode_records(tree, with_features=d > 0, with_edge_features=False)
    assert plain.numel() < buf.numel()
    eng.close()


def test_link_prediction_samples_with_edge_features():
    """pos_edges and the merged neighbourhood both carry Edge.feature_values (hydrateTaskBasedEdges,
    NodeAnchorBasedLinkPredictionBaseTask.scala:280-334)"""
    rng = np.random.default_rng(19)
    n, de = 3000, 5
    src, dst = rmat_edges(12, 20_000, seed=8)
    src, dst = (src % n).astype(np.uint32), (dst.astype(np.int64) * 11 % n).astype(np.uint32)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    feats = rng.standard_normal((n, 3)).astype(np.float32)
    f = _pair_features(de)
    eng = _engine(n, src, dst, feats)
    eng.load_edge_features(src, dst, np.stack([f(int(a), int(b)) for a, b in zip(src, dst)]), is_directed=False)
    roots = rng.integers(0, n, 48).astype(np.uint32)
    fanouts, P = [4, 3], 2
    pos, cnt = eng.sample_positives(roots, P)
    pos_h = pos.cpu().numpy().view(np.uint32).reshape(-1, P)
    cnt_h = cnt.cpu().numpy()
    all_roots = np.concatenate([roots[:, None], pos_h], axis=1).reshape(-1)
    tree = eng.sample_khop(all_roots, fanouts)
    buf, off = eng.encode_records(tree, kind=_lib.REC_NODE_ANCHOR_LINK_PRED, trees_per_record=1 + P,
                                  emit=(cnt > 0).to(torch.uint8))
    nbr = [t.cpu().numpy().view(np.uint32) for t in tree.nbr]
    lists = R.tree_edges(all_roots, fanouts, nbr)
    want = []
    for i, r in enumerate(roots.tolist()):
        if cnt_h[i] == 0:
            continue
        c = int(cnt_h[i])
        want.append(R.tfrecord_frame(R.nablp_sample_record(
            r, [lists[i * (1 + P) + k] for k in range(1 + c)], pos_h[i, :c].tolist(), c, feats, edge_features=f)))
    got = [g for g in _split(buf, off) if g]
    assert len(got) == len(want) and len(want) > 10
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, f"sample {i} differs"
    eng.close()


def test_user_defined_label_samples_match_host_assembly():
    """UserDefinedLabelsNodeAnchorBasedLinkPredictionTask: positives (counter 3) and hard negatives (counter 4) drawn
    from the user's edge lists; sample = root nbhd ++ positives' nbhds ++ negatives' nbhds; hard_neg_edges (= 2) sits
    between root_node and neighborhood on the wire; label edges carry the user tables' features (3 and 2 floats),
    neighbourhood edges the main table's (4 floats)"""
    rng = np.random.default_rng(23)
    n = 4000
    src, dst = rmat_edges(12, 30_000, seed=4)
    src, dst = (src % n).astype(np.uint32), (dst.astype(np.int64) * 13 % n).astype(np.uint32)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    feats = rng.standard_normal((n, 3)).astype(np.float32)
    f_main = _pair_features(4)
    eng = _engine(n, src, dst, feats)
    eng.load_edge_features(src, dst, np.stack([f_main(int(a), int(b)) for a, b in zip(src, dst)]), is_directed=False)
    # user-defined label edges: directed, with duplicates (first row's features win), some sources without negatives
    ps, pd = rng.integers(0, 400, 1500).astype(np.uint32), rng.integers(0, n, 1500).astype(np.uint32)
    ns, nd = rng.integers(0, 200, 500).astype(np.uint32), rng.integers(0, n, 500).astype(np.uint32)
    pf = rng.standard_normal((1500, 3)).astype(np.float32)
    nf = rng.standard_normal((500, 2)).astype(np.float32)
    eng.load_label_edges("pos", n, ps, pd, pf)
    eng.load_label_edges("neg", n, ns, nd, nf)
    first = lambda s_, d_, f_: {k: f_[i] for i, k in reversed(list(enumerate(zip(s_.tolist(), d_.tolist()))))}
    pos_feat, neg_feat = first(ps, pd, pf), first(ns, nd, nf)
    roots = np.concatenate([np.arange(0, 60), rng.integers(0, n, 20)]).astype(np.uint32)
    fanouts, P, Q = [4, 3], 2, 2
    pos, pcnt = eng.sample_positives(roots, P, label_edges="pos")
    neg, ncnt = eng.sample_positives(roots, Q, counter=4, label_edges="neg")
    pos_h, neg_h = pos.cpu().numpy().view(np.uint32).reshape(-1, P), neg.cpu().numpy().view(np.uint32).reshape(-1, Q)
    pc, nc = pcnt.cpu().numpy(), ncnt.cpu().numpy()
    # the label samples themselves: the oracle's permutation of the sorted distinct destinations
    for i, r in enumerate(roots.tolist()):
        for arr_s, arr_d, got, c, counter, f in ((ps, pd, pos_h[i], pc[i], 3, P), (ns, nd, neg_h[i], nc[i], 4, Q)):
            row = np.unique(arr_d[arr_s == r])
            want = np.sort(oracle.hash_permutation(row, r, sampling_seed=42, counter=counter)[:f]) if row.size else row
            assert c == want.size and np.array_equal(got[:c], want.astype(np.uint32))
    T = 1 + P + Q
    all_roots = np.concatenate([roots[:, None], pos_h, neg_h], axis=1).reshape(-1)
    tree = eng.sample_khop(all_roots, fanouts)
    buf, off = eng.encode_records(tree, kind=_lib.REC_NODE_ANCHOR_LINK_PRED, trees_per_record=T,
                                  emit=(pcnt > 0).to(torch.uint8), n_neg_trees=Q, pos_label_edges="pos",
                                  neg_label_edges="neg")
    nbr = [t.cpu().numpy().view(np.uint32) for t in tree.nbr]
    lists = R.tree_edges(all_roots, fanouts, nbr)
    want = []
    for i, r in enumerate(roots.tolist()):
        if pc[i] == 0:
            continue
        np_, nn_ = int(pc[i]), int(nc[i])
        trees = [lists[i * T]] + [lists[i * T + 1 + j] for j in range(np_)] + [lists[i * T + 1 + P + j] for j in range(nn_)]
        targets = pos_h[i, :np_].tolist() + neg_h[i, :nn_].tolist()
        want.append(R.tfrecord_frame(R.nablp_sample_record(
            r, trees, targets, np_, feats, edge_features=f_main, pos_edge_features=lambda a, b_: pos_feat[(a, b_)],
            neg_edge_features=lambda a, b_: neg_feat[(a, b_)])))
    got = [g for g in _split(buf, off) if g]
    assert len(got) == len(want) and len(want) > 30 and (nc > 0).sum() > 10 and ((pc > 0) & (nc == 0)).sum() > 5
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, f"sample {i} differs"
    eng.close()


def _random_engine(n, d, seed=5):
    rng = np.random.default_rng(seed)
    src, dst = rmat_edges(16, 400_000, seed=seed + 6)
    src, dst = (src.astype(np.int64) * 2654435761 % n).astype(np.uint32), (dst.astype(np.int64) * 40503 % n).astype(np.uint32)
    feats = rng.standard_normal((n, d)).astype(np.float32)
    return _engine(n, src, dst, feats), feats, rng


@pytest.mark.parametrize("fanouts,where", [([64, 64], "lds"), ([40, 30, 4], "scratch"), ([80, 5], "lds, fan-out > 64"),
                                           ([300], "lds, fan-out > 64"), ([3, 130], "lds, fan-out > 64"),
                                           ([100, 70], "scratch, fan-out > 64")])
def test_long_streams_beyond_the_old_2048_position_cap(fanouts, where):
    """4,161 stream positions still plan in LDS; 6,041 take the per-workgroup scratch plan (same code over global
    memory).  Both byte-identical to the restatement."""
    n = 60_000
    eng, feats, rng = _random_engine(n, 3)
    roots = rng.integers(0, n, 40).astype(np.uint32)
    tree = eng.sample_khop(roots, fanouts)
    buf, off = eng.encode_records(tree)
    nbr = [t.cpu().numpy().view(np.uint32) for t in tree.nbr]
    want = _host_rnn_frames(roots, fanouts, nbr, feats)
    got = _split(buf, off)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, f"record {i} differs ({where} plan): {len(g)} vs {len(w)} bytes"
    assert len(got) == len(want)
    eng.close()


def test_many_records_and_both_table_types():
    """many more records than waves in flight (the look-back chain runs over several rounds of tickets), fp32 and fp16
    tables: every frame's two CRC words against the restatement's CRC-32C (the payload's is assembled from the
    tabulated per-row states, gigl_features_row_crc, and the header bytes folded while they were written)"""
    n = 50_000
    for dtype in (torch.float32, torch.float16):
        eng, feats, rng = _random_engine(n, 36)
        if dtype is torch.float16:
            eng.load_features(torch.from_numpy(feats).to(torch.float16))
            feats = feats.astype(np.float16).astype(np.float32)
        roots = rng.integers(0, n, 40_000).astype(np.uint32)
        fanouts = [5, 3]
        tree = eng.sample_khop(roots, fanouts)
        buf, off = eng.encode_records(tree)
        frames = _split(buf, off)
        assert len(frames) == 40_000
        for f in frames[:50] + frames[-50:] + frames[20_000:20_050]:
            assert R.tfrecord_frame(f[12:-4]) == f
        nbr = [t.cpu().numpy().view(np.uint32) for t in tree.nbr]
        idx = np.concatenate([np.arange(40), np.arange(39_960, 40_000)])
        want = _host_rnn_frames(roots[idx], fanouts, [nb.reshape(40_000, -1)[idx].reshape(-1) for nb in nbr], feats)
        assert [frames[i] for i in idx.tolist()] == want
        eng.close()


def test_wave_per_record_plans_against_the_restatement_frame_by_frame():
    """calls of more than 6,144 records plan with ONE WAVE per record (build_plan<64>: its phases are ordered by
    wavefront fences, not workgroup barriers) — every frame of a 7,000-record call, and of a 7,000-record call whose
    records are too long for LDS (the scratch plan of the same code), against the restatement"""
    n = 50_000
    for fanouts, nrec in (([5, 3], 7_000), ([40, 30, 4], 300)):
        eng, feats, rng = _random_engine(n, 5)
        roots = rng.integers(0, n, nrec).astype(np.uint32)
        tree = eng.sample_khop(roots, fanouts)
        buf, off = eng.encode_records(tree)
        nbr = [t.cpu().numpy().view(np.uint32) for t in tree.nbr]
        want = _host_rnn_frames(roots, fanouts, nbr, feats)
        got = _split(buf, off)
        assert len(got) == len(want) == nrec
        bad = [i for i, (g, w) in enumerate(zip(got, want)) if g != w]
        assert not bad, f"{len(bad)} of {nrec} records differ (first {bad[:5]}), fanouts {fanouts}"
        eng.close()


def test_row_crc_table_matches_the_restatement():
    import ctypes as C
    n, d = 1000, 37
    eng, feats, _ = _random_engine(n, d)
    tbl = C.c_void_p()
    _lib.check(eng._lib.gigl_features_row_crc(eng._ctx, eng._feat, C.byref(tbl)), eng._ctx)
    got = np.empty(n, dtype=np.uint32)
    _lib.check(eng._lib.gigl_memcpy(eng._ctx, got.ctypes.data_as(C.c_void_p), _lib.LOC_HOST, tbl, _lib.LOC_DEVICE, n * 4),
               eng._ctx)

    def raw(data):  # CRC-32C register after `data` from a zero start, no final inversion
        R.crc32c(b"")
        c = 0
        for b in data:
            c = R._CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
        return c
    for i in (0, 1, 17, n - 1):
        assert int(got[i]) == raw(feats[i].astype("<f4").tobytes())
    eng.close()


def test_output_capacity_too_small_is_reported():
    import ctypes as C
    n = 20_000
    eng, feats, rng = _random_engine(n, 8)
    roots = rng.integers(0, n, 500).astype(np.uint32)
    tree = eng.sample_khop(roots, [4, 3])
    buf, off = eng.encode_records(tree)
    need = int(off[-1].item())
    o = _lib.GiglRecordOpts()
    o.kind, o.trees_per_record, o.tfrecord_frame = _lib.REC_ROOTED_NODE_NEIGHBORHOOD, 1, 1
    cap = need - 1
    out = torch.zeros(need + 4096, dtype=torch.uint8, device="cuda")
    rec_off = torch.empty(501, dtype=torch.int64, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(eng._lib.gigl_records_encode(eng._ctx, C.c_void_p(tree.roots.data_ptr()), C.byref(tree.c_struct), eng._feat,
                                            C.byref(o), 500, C.c_void_p(out.data_ptr()), cap,
                                            C.c_void_p(rec_off.data_ptr()), C.c_void_p(status.data_ptr())), eng._ctx)
    eng._stream.synchronize()
    assert int(status.item()) == 1
    assert int(out[cap:].sum().item()) == 0  # nothing written beyond the capacity it was given
    assert torch.equal(rec_off.cpu(), off.cpu())  # the offsets (and so the size to retry with) are still exact
    eng.close()


def test_oversized_record_is_rejected():
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    rowptr = np.zeros(11, dtype=np.int64)
    eng.load_csc(rowptr, np.zeros(0, dtype=np.uint32))
    tree = eng.sample_khop(np.arange(2, dtype=np.uint32), [64, 64, 64, 5])  # 1.58 M slots per tree > 2^20
    with pytest.raises(_lib.GiglError):
        eng.encode_records(tree, with_features=False)
    eng.close()
