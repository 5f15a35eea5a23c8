"""Pin of oracle/records.py (the restatement the device record encoder is checked against) on reference-held data:
encoding the decoded contents of the reference's own sampler output fixtures — decoded by the reference's generated
protobuf code into tests/golden/ref_assets_decoded.json — reproduces the fixture FILES byte for byte, TFRecord length
and payload CRCs included.  CPU only."""
import json
import os
import struct

import numpy as np
import pytest

from oracle import records as R

A = "ref_assets"
RNN = "split_generator/supervised_node_classification/sgs_output/unlabeled/samples/data.tfrecord"
SNC = "split_generator/supervised_node_classification/sgs_output/labeled/samples/data.tfrecord"
NABLP = "split_generator/node_anchor_based_link_prediction/sgs_output/node_anchor_based_link_prediction_samples/data.tfrecord"
RNEG = "split_generator/node_anchor_based_link_prediction/sgs_output/random_negative_rooted_neighborhood_samples/user/data.tfrecord"


@pytest.fixture(scope="module")
def decoded(golden_dir):
    return json.load(open(os.path.join(golden_dir, "ref_assets_decoded.json")))


def _node(d):
    return R.encode_node(d["node_id"], d.get("condensed_node_type"), d["feature_values"])


def _edge(d):
    return R.encode_edge(d["src_node_id"], d["dst_node_id"], d.get("condensed_edge_type"), d["feature_values"])


def _graph(d):
    return R.encode_graph([_node(n) for n in d["nodes"]], [_edge(e) for e in d["edges"]])


def test_crc32c_and_varint_known_answers():
    assert R.crc32c(b"123456789") == 0xE3069283 and R.crc32c(bytes(range(32))) == 0x46DD794E  # RFC 3720 B.4
    assert R._varint(300) == b"\xac\x02" and R._varint(-1) == b"\xff" * 9 + b"\x01"  # protobuf encoding guide


@pytest.mark.parametrize("rel", [RNN, RNEG, SNC, NABLP])
def test_reference_sampler_outputs_are_reproduced_byte_for_byte(golden_dir, decoded, rel):
    want = open(os.path.join(golden_dir, A, rel), "rb").read()
    frames = []
    for d in decoded[rel]["records"]:
        root, graph = _node(d["root_node"]), _graph(d["neighborhood"])
        if rel == SNC:
            payload = R.encode_supervised_node_classification_sample(
                root, graph, [R.encode_label(lb["label_type"], lb["label"]) for lb in d["root_node_labels"]])
        elif rel == NABLP:
            payload = R.encode_nablp_sample(root, [_edge(e) for e in d["hard_neg_edges"]],
                                            [_edge(e) for e in d["pos_edges"]], graph, [_edge(e) for e in d["neg_edges"]])
        else:
            payload = R.encode_rooted_node_neighborhood(root, graph)
        assert payload.hex() == d["reserialized_hex"]  # == the reference's generated code re-serialising the record
        frames.append(R.tfrecord_frame(payload))
    assert b"".join(frames) == want  # the reference's file: framing, both CRCs of every record


def test_assembly_rule_on_a_hand_made_tree():
    """createSubgraph on a two-hop tree: edges hop 1 then hop 2, nodes = distinct sources in edge order + the root; a
    root without in-edges yields nodes = [root], no edges (createIsolatedNodesSubgraph)"""
    roots = np.array([5, 9], dtype=np.uint32)
    inv = R.INVALID
    nbr0 = np.array([1, 2, 0 + inv, inv], dtype=np.uint32)          # root 5 <- {1, 2}; root 9 <- {}
    nbr1 = np.array([2, 7, 5, inv, inv, inv, inv, inv], dtype=np.uint32)  # 1 <- {2, 7}; 2 <- {5}
    (s0, d0), (s1, d1) = R.tree_edges(roots, [2, 2], [nbr0, nbr1])
    assert s0.tolist() == [1, 2, 2, 7, 5] and d0.tolist() == [5, 5, 1, 1, 2] and s1.size == 0
    order, nodes, edges = R.assemble(5, s0, d0, None)
    assert order == [1, 2, 7, 5] and [(a, b) for a, b, _ in edges] == [(1, 5), (2, 5), (2, 1), (7, 1), (5, 2)]
    order, nodes, edges = R.assemble(9, s1, d1, None)
    assert order == [9] and not edges
