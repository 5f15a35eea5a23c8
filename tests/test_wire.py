"""gigl_amd.wire against the reference's own fixture files and the reference's generated-code decode."""
import json
import os

import numpy as np
import pytest

from gigl_amd import wire

A = "ref_assets"
RNN = "split_generator/supervised_node_classification/sgs_output/unlabeled/samples/data.tfrecord"
SNC = "split_generator/supervised_node_classification/sgs_output/labeled/samples/data.tfrecord"
NABLP = "split_generator/node_anchor_based_link_prediction/sgs_output/node_anchor_based_link_prediction_samples/data.tfrecord"
RNEG = "split_generator/node_anchor_based_link_prediction/sgs_output/random_negative_rooted_neighborhood_samples/user/data.tfrecord"


@pytest.fixture(scope="module")
def decoded(golden_dir):
    return json.load(open(os.path.join(golden_dir, "ref_assets_decoded.json")))


def _node_eq(n, d):
    assert n.node_id == d["node_id"]
    assert n.condensed_node_type == d.get("condensed_node_type")
    assert np.array_equal(n.feature_values, np.asarray(d["feature_values"], dtype=np.float32))


def _edge_eq(e, d):
    assert (e.src_node_id, e.dst_node_id) == (d["src_node_id"], d["dst_node_id"])
    assert e.condensed_edge_type == d.get("condensed_edge_type")
    assert np.array_equal(e.feature_values, np.asarray(d["feature_values"], dtype=np.float32))


def _graph_eq(g, d):
    assert len(g.nodes) == len(d["nodes"]) and len(g.edges) == len(d["edges"])
    for n, dn in zip(g.nodes, d["nodes"]):
        _node_eq(n, dn)
    for e, de in zip(g.edges, d["edges"]):
        _edge_eq(e, de)


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert wire.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert wire.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert wire.crc32c(bytes(range(32))) == 0x46DD794E
    assert wire.crc32c(b"123456789") == 0xE3069283


@pytest.mark.parametrize("rel,cls", [(RNN, wire.RootedNodeNeighborhood), (RNEG, wire.RootedNodeNeighborhood)])
def test_rnn_fixture_decode_and_roundtrip(golden_dir, decoded, rel, cls):
    path = os.path.join(golden_dir, A, rel)
    recs = list(wire.read_tfrecords(path, verify_crc=True))  # CRCs of the reference writer verify
    exp = decoded[rel]["records"]
    assert len(recs) == len(exp) == 16
    for raw, d in zip(recs, exp):
        m = cls.FromString(raw)
        _node_eq(m.root_node, d["root_node"])
        _graph_eq(m.neighborhood, d["neighborhood"])
        # byte-exact: our encoder == ScalaPB's bytes in the file == protobuf-python's re-encode
        assert m.SerializeToString() == raw
        assert m.SerializeToString().hex() == d["reserialized_hex"]


def test_snc_fixture(golden_dir, decoded):
    recs = list(wire.read_tfrecords(os.path.join(golden_dir, A, SNC)))
    exp = decoded[SNC]["records"]
    assert len(recs) == len(exp) == 14
    for raw, d in zip(recs, exp):
        m = wire.SupervisedNodeClassificationSample.FromString(raw)
        _node_eq(m.root_node, d["root_node"])
        _graph_eq(m.neighborhood, d["neighborhood"])
        assert [(l.label_type, l.label) for l in m.root_node_labels] == [
            (x["label_type"], x["label"]) for x in d["root_node_labels"]]
        assert m.SerializeToString() == raw


def test_nablp_fixture(golden_dir, decoded):
    recs = list(wire.read_tfrecords(os.path.join(golden_dir, A, NABLP)))
    exp = decoded[NABLP]["records"]
    assert len(recs) == len(exp) == 14
    for raw, d in zip(recs, exp):
        m = wire.NodeAnchorBasedLinkPredictionSample.FromString(raw)
        _node_eq(m.root_node, d["root_node"])
        _graph_eq(m.neighborhood, d["neighborhood"])
        for k in ("pos_edges", "hard_neg_edges", "neg_edges"):
            assert len(getattr(m, k)) == len(d[k])
            for e, de in zip(getattr(m, k), d[k]):
                _edge_eq(e, de)
        assert m.SerializeToString().hex() == d["reserialized_hex"]


def test_tfrecord_file_roundtrip_bytes(golden_dir, tmp_path):
    src = os.path.join(golden_dir, A, RNN)
    recs = list(wire.read_tfrecords(src))
    out = tmp_path / "x.tfrecord"
    wire.write_tfrecords(str(out), recs)
    assert out.read_bytes() == open(src, "rb").read()


def test_tfrecord_corruption_detected(golden_dir):
    data = bytearray(open(os.path.join(golden_dir, A, RNN), "rb").read())
    data[20] ^= 0x01
    with pytest.raises(ValueError):
        list(wire.iter_tfrecords(bytes(data)))
    with pytest.raises(ValueError):
        list(wire.iter_tfrecords(bytes(data[:-3])))


def test_tf_example_inputs(golden_dir):
    nodes = [wire.decode_tf_example(r) for r in wire.read_tfrecords(os.path.join(
        golden_dir, A, "subgraph_sampler/supervised_node_classification/node_data/data.tfrecord"))]
    edges = [wire.decode_tf_example(r) for r in wire.read_tfrecords(os.path.join(
        golden_dir, A, "subgraph_sampler/supervised_node_classification/edge_data/data.tfrecord"))]
    assert len(nodes) == 16 and len(edges) == 34
    assert sorted(int(n["node_id"][0]) for n in nodes) == list(range(16))
    assert set(nodes[0]) >= {"node_id", "f0", "f1"}
    for e in edges:
        assert 0 <= int(e["src"][0]) < 16 and 0 <= int(e["dst"][0]) < 16
    # encode -> decode is the identity
    back = wire.decode_tf_example(wire.encode_tf_example(nodes[3]))
    assert set(back) == set(nodes[3])
    for k in back:
        assert np.array_equal(np.asarray(back[k]), np.asarray(nodes[3][k]))


def test_default_elision_and_presence():
    n = wire.Node(node_id=0, condensed_node_type=0)
    assert n.SerializeToString() == b"\x10\x00"  # id 0 elided, optional type 0 present
    assert wire.Node.FromString(b"\x10\x00") == n
    assert wire.Node.FromString(b"") == wire.Node()
    e = wire.Edge(src_node_id=300, dst_node_id=0, feature_values=np.array([1.5], np.float32))
    assert wire.Edge.FromString(e.SerializeToString()) == e
    lb = wire.Label(label_type="node_label", label=-3)
    assert wire.Label.FromString(lb.SerializeToString()) == lb
