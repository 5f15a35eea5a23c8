"""ORACLE (test infrastructure only — nothing under gigl_amd/ may import this): CPU restatement of the sampler job's
OUTPUT stage — per-root assembly of a sampled tree into RootedNodeNeighborhood /
NodeAnchorBasedLinkPredictionSample messages, proto3 wire encoding and TFRecord framing — against which the device
encoder (gigl_records_encode) is checked byte for byte (tests/test_gpu_records.py).

Follows (paths relative to the reference root):
  createSubgraph            scala/subgraph_sampler/.../pureSpark/SGSPureSparkV1Task.scala:671-820
      edges = hop-1 edges ++ hop-2 edges ++ ...; nodes = array_distinct(hop-1 nodes ++ hop-2 nodes ++ [root]);
      every node / edge carries its condensed type and its hydrated feature values (:496-593)
  createIsolated / Neighborless   :847-971   a root without in-edges: nodes = [root], no edges
  castToRootedNodeNeighborhoodProtoSchema :1019-1040, castToTrainingSampleProtoSchema
      (.../NodeAnchorBasedLinkPredictionBaseTask.scala:388-426)
  createNodeAnchorBasedLinkPredictionSubgraph  .../NodeAnchorBasedLinkPredictionTask.scala:146-312
      neighborhood = array_distinct(root nbhd ++ the positives' (and hard negatives') nbhds), first occurrence wins;
      pos_edges = [root -> positive], hard_neg_edges = [root -> negative] (user-defined labels only)
  message layouts           proto/snapchat/research/gbml/graph_schema.proto:5-40,
                            proto/snapchat/research/gbml/training_samples_schema.proto:7-41
  TFRecordIO.writeDatasetToTfrecord  scala/common/src/main/scala/utils/TFRecordIO.scala:53-69 (TFRecord framing:
      u64 length, masked CRC-32C of the length, payload, masked CRC-32C of the payload)

Pinned on reference-held data: `encode_*` of the decoded contents of the reference's own sampler output fixtures
(tests/golden/ref_assets_decoded.json, produced by the reference's generated protobuf code) reproduces the fixture
files byte for byte, TFRecord CRCs included (tests/test_oracle_records.py).  The ORDER of nodes inside a record is not
defined by the reference (Spark's collect_list order); the device and this restatement use: sources in edge order
(hop 1 first), the root last if it was not seen — a fixed representative of the reference's set.
"""
from __future__ import annotations

import struct
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

INVALID = 0xFFFFFFFF

# ---- proto3 wire format (only what these messages use) ----------------------------------------------------------------


def _varint(v: int) -> bytes:
    v &= 0xFFFFFFFFFFFFFFFF  # negative int32 / int64 values are written as 10-byte two's complement varints
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _key(field: int, wire_type: int) -> bytes:
    return _varint((field << 3) | wire_type)


def _ld(field: int, payload: bytes) -> bytes:
    return _key(field, 2) + _varint(len(payload)) + payload


def _floats(field: int, values) -> bytes:
    a = np.asarray(values, dtype="<f4")
    return _ld(field, a.tobytes()) if a.size else b""  # packed repeated float; an empty list writes nothing


def encode_node(node_id: int, condensed_node_type: Optional[int], feature_values) -> bytes:
    """Node {uint32 node_id = 1; optional uint32 condensed_node_type = 2; repeated float feature_values = 3}"""
    out = b""
    if node_id:  # proto3: a zero scalar without presence is not written
        out += _key(1, 0) + _varint(node_id)
    if condensed_node_type is not None:  # `optional`: presence is explicit, 0 is written
        out += _key(2, 0) + _varint(condensed_node_type)
    return out + _floats(3, feature_values)


def encode_edge(src: int, dst: int, condensed_edge_type: Optional[int], feature_values) -> bytes:
    """Edge {uint32 src_node_id = 1; uint32 dst_node_id = 2; optional uint32 condensed_edge_type = 3;
    repeated float feature_values = 4}"""
    out = b""
    if src:
        out += _key(1, 0) + _varint(src)
    if dst:
        out += _key(2, 0) + _varint(dst)
    if condensed_edge_type is not None:
        out += _key(3, 0) + _varint(condensed_edge_type)
    return out + _floats(4, feature_values)


def encode_graph(nodes: Sequence[bytes], edges: Sequence[bytes]) -> bytes:
    """Graph {repeated Node nodes = 2; repeated Edge edges = 3} (graph_schema.proto:59-62) from already encoded members"""
    return b"".join(_ld(2, n) for n in nodes) + b"".join(_ld(3, e) for e in edges)


def encode_rooted_node_neighborhood(root: bytes, graph: bytes) -> bytes:
    """RootedNodeNeighborhood {Node root_node = 1; Graph neighborhood = 2}"""
    return _ld(1, root) + _ld(2, graph)


def encode_label(label_type: str, label: int) -> bytes:
    """Label {string label_type = 1; int32 label = 2}"""
    out = _ld(1, label_type.encode("utf-8")) if label_type else b""
    if label:
        out += _key(2, 0) + _varint(label)
    return out


def encode_supervised_node_classification_sample(root: bytes, graph: bytes, labels: Sequence[bytes]) -> bytes:
    """SupervisedNodeClassificationSample {Node root_node = 1; Graph neighborhood = 2; repeated Label root_node_labels = 3}"""
    return _ld(1, root) + _ld(2, graph) + b"".join(_ld(3, lb) for lb in labels)


def encode_nablp_sample(root: bytes, hard_neg_edges: Sequence[bytes], pos_edges: Sequence[bytes], graph: bytes,
                        neg_edges: Sequence[bytes] = ()) -> bytes:
    """NodeAnchorBasedLinkPredictionSample {Node root_node = 1; repeated Edge hard_neg_edges = 2; Graph neighborhood = 3;
    repeated Edge pos_edges = 4; repeated Edge neg_edges = 5} — fields in field-number order"""
    return (_ld(1, root) + b"".join(_ld(2, e) for e in hard_neg_edges) + _ld(3, graph) +
            b"".join(_ld(4, e) for e in pos_edges) + b"".join(_ld(5, e) for e in neg_edges))


# ---- TFRecord framing ---------------------------------------------------------------------------------------------

_CRC_TABLE: List[int] = []


def crc32c(data: bytes) -> int:
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), bitwise-table form"""
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def tfrecord_frame(payload: bytes) -> bytes:
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", _masked(crc32c(head))) + payload + struct.pack("<I", _masked(crc32c(payload)))


# ---- per-root assembly (createSubgraph) ---------------------------------------------------------------------------


def tree_edges(roots: np.ndarray, fanouts: Sequence[int], nbr: Sequence[np.ndarray]) -> List[Tuple[np.ndarray, np.ndarray]]:
    """per root the (src, dst) arrays of its sampled tree in the order hop-1 edges, hop-2 edges, ...; slot p*f + j of
    hop k is the j-th sample under parent slot p (include/gigl_hip.h tree layout), INVALID = empty"""
    roots = np.asarray(roots, dtype=np.uint32)
    b = roots.size
    out_s: List[List[int]] = [[] for _ in range(b)]
    out_d: List[List[int]] = [[] for _ in range(b)]
    parents = roots.astype(np.int64)
    per_root = 1
    for k, f in enumerate(int(v) for v in fanouts):
        a = np.asarray(nbr[k], dtype=np.uint32).astype(np.int64).reshape(b, per_root * f)
        par = parents.reshape(b, per_root)
        for i in range(b):
            for p in range(per_root):
                for j in range(f):
                    s = int(a[i, p * f + j])
                    if s != INVALID:
                        out_s[i].append(s)
                        out_d[i].append(int(par[i, p]))
        parents = a.reshape(-1)
        per_root *= f
    return [(np.array(s, dtype=np.int64), np.array(d, dtype=np.int64)) for s, d in zip(out_s, out_d)]


def assemble(root: int, src: np.ndarray, dst: np.ndarray, features: Optional[np.ndarray],
             condensed_node_type: Optional[int] = 0, condensed_edge_type: Optional[int] = 0,
             edge_features: Optional[Callable[[int, int], np.ndarray]] = None):
    """one root's (node id list, {node id: encoded Node}, [(src, dst, encoded Edge)]): distinct nodes in first-seen order
    over the edge sources, the root last if unseen; the edges as sampled (a set on simple graphs)"""
    order: List[int] = []
    seen = set()
    for v in list(src.tolist()) + [int(root)]:
        if v not in seen:
            seen.add(v)
            order.append(v)
    fv = (lambda v: features[v]) if features is not None else (lambda v: ())
    nodes = {v: encode_node(v, condensed_node_type, fv(v)) for v in order}
    edges = [(int(s), int(d), encode_edge(int(s), int(d), condensed_edge_type,
                                          edge_features(int(s), int(d)) if edge_features else ()))
             for s, d in zip(src.tolist(), dst.tolist())]
    return order, nodes, edges


def rooted_node_neighborhood_record(root: int, src, dst, features, condensed_node_type=0, condensed_edge_type=0,
                                    edge_features=None, suffix: bytes = b"") -> bytes:
    """the RootedNodeNeighborhood payload of one root (+ `suffix`: the encoded root_node_labels of a
    SupervisedNodeClassificationSample, which shares fields 1 and 2)"""
    order, nodes, edges = assemble(root, np.asarray(src), np.asarray(dst), features, condensed_node_type,
                                   condensed_edge_type, edge_features)
    fv = features[int(root)] if features is not None else ()
    return encode_rooted_node_neighborhood(encode_node(int(root), condensed_node_type, fv),
                                           encode_graph([nodes[v] for v in order], [e for _, _, e in edges])) + suffix


def nablp_sample_record(root: int, trees: Sequence[Tuple[np.ndarray, np.ndarray]], targets: Sequence[int],
                        n_pos: int, features, condensed_node_type=0, condensed_edge_type=0, edge_features=None,
                        pos_edge_features=None, neg_edge_features=None) -> bytes:
    """trees[0] = the root's own tree, trees[1 + j] = the tree of targets[j]; the first n_pos targets are positives, the
    rest hard negatives.  neighbourhood = first-occurrence union of all trees' nodes and edges"""
    node_order: List[int] = []
    node_bytes: Dict[int, bytes] = {}
    edge_bytes: Dict[Tuple[int, int], bytes] = {}
    for t, (s, d) in zip([int(root)] + [int(v) for v in targets], trees):
        order, nodes, edges = assemble(t, np.asarray(s), np.asarray(d), features, condensed_node_type, condensed_edge_type,
                                       edge_features)
        for v in order:
            if v not in node_bytes:
                node_bytes[v] = nodes[v]
                node_order.append(v)
        for a, b_, e in edges:
            edge_bytes.setdefault((a, b_), e)
    lf_pos = pos_edge_features or edge_features
    lf_neg = neg_edge_features or edge_features
    pos = [encode_edge(int(root), int(t), condensed_edge_type, lf_pos(int(root), int(t)) if lf_pos else ())
           for t in targets[:n_pos]]
    neg = [encode_edge(int(root), int(t), condensed_edge_type, lf_neg(int(root), int(t)) if lf_neg else ())
           for t in targets[n_pos:]]
    fv = features[int(root)] if features is not None else ()
    return encode_nablp_sample(encode_node(int(root), condensed_node_type, fv), neg, pos,
                               encode_graph([node_bytes[v] for v in node_order], list(edge_bytes.values())))
