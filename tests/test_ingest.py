"""native TFRecord / tf.Example reader (host C++ in libgigl_hip.so) vs the Python codec gigl_amd/wire.py, on the
reference's own sampler input fixtures and on synthetic tables (packed / unpacked lists, missing keys, negative
ids, integer feature columns, corrupt files).  Host code only: runs without a GPU."""
import os

import numpy as np
import pytest

from gigl_amd import wire
from gigl_amd.ingest import COL_F32, COL_I64, read_columns

A = "ref_assets/subgraph_sampler/supervised_node_classification"


def test_reference_node_and_edge_tables(golden_dir):
    nd = os.path.join(golden_dir, A, "node_data/data.tfrecord")
    ed = os.path.join(golden_dir, A, "edge_data/data.tfrecord")
    want = [wire.decode_tf_example(r) for r in wire.read_tfrecords(nd)]
    got, cnt = read_columns([nd], [("node_id", COL_I64, 1), ("f0", COL_F32, 1), ("f1", COL_F32, 1),
                                   ("node_label", COL_I64, 1)])
    assert got["node_id"][:, 0].tolist() == [int(r["node_id"][0]) for r in want]
    assert np.array_equal(got["f0"][:, 0], np.array([np.asarray(r["f0"], np.float32)[0] for r in want]))
    assert np.array_equal(got["f1"][:, 0], np.array([np.asarray(r["f1"], np.float32)[0] for r in want]))
    assert got["node_label"][:, 0].tolist() == [int(r["node_label"][0]) for r in want]
    assert (cnt["node_id"] == 1).all()
    want_e = [wire.decode_tf_example(r) for r in wire.read_tfrecords(ed)]
    got_e, _ = read_columns([ed], [("src", COL_I64, 1), ("dst", COL_I64, 1)])
    assert got_e["src"][:, 0].tolist() == [int(r["src"][0]) for r in want_e]
    assert got_e["dst"][:, 0].tolist() == [int(r["dst"][0]) for r in want_e]
    assert len(want_e) == 34 and len(want) == 16


def _unpacked_int_feature(values):
    """Feature{int64_list{value: v, value: v ...}} with one tag per element (writers may choose either form)"""
    lst = b"".join(wire._tag(1, 0) + wire._enc_varint(v) for v in values)
    return wire._len_delim(3, lst)


def test_synthetic_table_multithreaded(tmp_path):
    rng = np.random.default_rng(0)
    n = 20_000
    ids = rng.permutation(n).astype(np.int64)
    ids[7] = -3  # int64 two's complement (10-byte varint)
    feats = rng.standard_normal((n, 5)).astype(np.float32)
    ints = rng.integers(-50, 50, (n, 2))
    payloads = []
    for i in range(n):
        f = {"id": np.array([ids[i]]), "x": feats[i], "cat": ints[i]}
        if i % 3 == 0:
            f["label"] = np.array([i % 7])
        payloads.append(wire.encode_tf_example(f))
    # record 11: the same content with an unpacked int64 list for "cat"
    ex = wire.decode_tf_example(payloads[11])
    entries = b""
    for key in sorted(ex):
        if key == "cat":
            feat = _unpacked_int_feature([int(v) for v in ints[11]])
        elif key == "x":
            feat = wire._len_delim(2, wire._len_delim(1, feats[11].tobytes()))
        else:
            feat = wire._len_delim(3, wire._len_delim(1, b"".join(wire._enc_varint(int(v)) for v in np.atleast_1d(ex[key]))))
        entries += wire._len_delim(1, wire._len_delim(1, key.encode()) + wire._len_delim(2, feat))
    payloads[11] = wire._len_delim(1, entries)
    p1, p2 = str(tmp_path / "a.tfrecord"), str(tmp_path / "b.tfrecord")
    wire.write_tfrecords(p1, payloads[:12_000])
    wire.write_tfrecords(p2, payloads[12_000:])
    got, cnt = read_columns([p1, p2], [("id", COL_I64, 1), ("x", COL_F32, 5), ("cat", COL_F32, 2), ("label", COL_I64, 1),
                                       ("absent", COL_F32, 3)], n_threads=4)
    assert np.array_equal(got["id"][:, 0], ids)
    assert np.array_equal(got["x"], feats)
    assert np.array_equal(got["cat"], ints.astype(np.float32))  # integer feature column cast to float
    assert np.array_equal(cnt["label"], (np.arange(n) % 3 == 0).astype(np.int32))
    assert np.array_equal(got["label"][::3, 0], np.arange(n)[::3] % 7)
    assert (got["absent"] == 0).all() and (cnt["absent"] == 0).all()


def test_corrupt_files_are_rejected(tmp_path):
    p = str(tmp_path / "t.tfrecord")
    wire.write_tfrecords(p, [wire.encode_tf_example({"id": np.array([1])}) for _ in range(10)])
    raw = bytearray(open(p, "rb").read())
    bad = bytearray(raw)
    bad[40] ^= 0x10
    open(p, "wb").write(bad)
    with pytest.raises(ValueError):
        read_columns([p], [("id", COL_I64, 1)])
    open(p, "wb").write(raw[:-3])  # truncated
    with pytest.raises(ValueError):
        read_columns([p], [("id", COL_I64, 1)])
    open(p, "wb").write(raw)
    with pytest.raises(ValueError):  # wrong kind: a float column asked as int64
        wire.write_tfrecords(p, [wire.encode_tf_example({"x": np.array([1.5], np.float32)})])
        read_columns([p], [("x", COL_I64, 1)])
