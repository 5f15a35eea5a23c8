"""Byte-compatible protobuf + TFRecord codec for the data (wire) boundary of the hot path.

The reference exchanges samples between its Spark sampler and its trainer/inferencer as TFRecord
files of serialized protos (SURVEY.md §8(b) "Data (wire) boundary").  `protoc` is not available in
this image and the reference's generated `*_pb2.py` never travels with this repo, so the six small
messages on the path are hand-coded here against the schema:

  proto/snapchat/research/gbml/graph_schema.proto:5-62        Node, Edge, Graph
  proto/snapchat/research/gbml/training_samples_schema.proto:9-53
        Label, RootedNodeNeighborhood, SupervisedNodeClassificationSample,
        NodeAnchorBasedLinkPredictionSample
  TFRecord framing: scala/common/src/main/scala/utils/TFRecordIO.scala:53-69 (spark-tfrecord
        "ByteArray" record type) == TensorFlow's record format
        (u64 length | masked crc32c(length) | payload | masked crc32c(payload)).
  tf.Example (input node/edge tables): read by SGSPureSparkV1Task.scala:52-118,120-286.

Method names follow the protobuf Python API (`SerializeToString`, `FromString`) so code written
against the reference's generated classes reads the same.  Encoding choices match ScalaPB /
protobuf-python output byte-for-byte (pinned by tests/test_wire.py on the reference's own fixtures):
fields in field-number order, proto3 default elision for non-optional scalars, explicit presence
for `optional` fields, packed repeated floats.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

# ------------------------------------------------------------------------------------------------
# varint / wire primitives
# ------------------------------------------------------------------------------------------------


def _enc_varint(v: int) -> bytes:
    v &= 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _dec_varint(buf, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _tag(field_no: int, wire_type: int) -> bytes:
    return _enc_varint((field_no << 3) | wire_type)


def _len_delim(field_no: int, payload: bytes) -> bytes:
    return _tag(field_no, 2) + _enc_varint(len(payload)) + payload


def _iter_fields(buf) -> Iterator[Tuple[int, int, object]]:
    """yields (field_no, wire_type, value) — value is int for varint/fixed, memoryview for len-delim"""
    mv = memoryview(buf)
    pos, n = 0, len(mv)
    while pos < n:
        key, pos = _dec_varint(mv, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _dec_varint(mv, pos)
            yield fno, wt, v
        elif wt == 1:
            yield fno, wt, bytes(mv[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _dec_varint(mv, pos)
            if pos + ln > n:
                raise ValueError("truncated length-delimited field")
            yield fno, wt, mv[pos:pos + ln]
            pos += ln
        elif wt == 5:
            yield fno, wt, bytes(mv[pos:pos + 4])
            pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")


def _floats_from(wt: int, v, acc: List[np.ndarray]) -> None:
    if wt == 2:  # packed
        acc.append(np.frombuffer(bytes(v), dtype="<f4"))
    elif wt == 5:  # unpacked element
        acc.append(np.frombuffer(v, dtype="<f4"))
    else:
        raise ValueError("bad wire type for repeated float")


_EMPTY_F32 = np.zeros(0, dtype=np.float32)


def _cat(acc: List[np.ndarray]) -> np.ndarray:
    if not acc:
        return _EMPTY_F32
    return acc[0] if len(acc) == 1 else np.concatenate(acc)


# ------------------------------------------------------------------------------------------------
# graph_schema.proto
# ------------------------------------------------------------------------------------------------


@dataclass
class Node:
    """graph_schema.proto:5-12"""
    node_id: int = 0
    condensed_node_type: Optional[int] = None
    feature_values: np.ndarray = field(default_factory=lambda: _EMPTY_F32)

    def SerializeToString(self) -> bytes:
        out = bytearray()
        if self.node_id:
            out += _tag(1, 0) + _enc_varint(self.node_id)
        if self.condensed_node_type is not None:
            out += _tag(2, 0) + _enc_varint(self.condensed_node_type)
        fv = np.asarray(self.feature_values, dtype="<f4")
        if fv.size:
            out += _len_delim(3, fv.tobytes())
        return bytes(out)

    @classmethod
    def FromString(cls, buf) -> "Node":
        m = cls()
        acc: List[np.ndarray] = []
        for fno, wt, v in _iter_fields(buf):
            if fno == 1:
                m.node_id = int(v) & 0xFFFFFFFF
            elif fno == 2:
                m.condensed_node_type = int(v) & 0xFFFFFFFF
            elif fno == 3:
                _floats_from(wt, v, acc)
        m.feature_values = _cat(acc)
        return m

    def __eq__(self, o):
        return (isinstance(o, Node) and self.node_id == o.node_id
                and self.condensed_node_type == o.condensed_node_type
                and np.array_equal(self.feature_values, o.feature_values))


@dataclass
class Edge:
    """graph_schema.proto:16-25"""
    src_node_id: int = 0
    dst_node_id: int = 0
    condensed_edge_type: Optional[int] = None
    feature_values: np.ndarray = field(default_factory=lambda: _EMPTY_F32)

    def SerializeToString(self) -> bytes:
        out = bytearray()
        if self.src_node_id:
            out += _tag(1, 0) + _enc_varint(self.src_node_id)
        if self.dst_node_id:
            out += _tag(2, 0) + _enc_varint(self.dst_node_id)
        if self.condensed_edge_type is not None:
            out += _tag(3, 0) + _enc_varint(self.condensed_edge_type)
        fv = np.asarray(self.feature_values, dtype="<f4")
        if fv.size:
            out += _len_delim(4, fv.tobytes())
        return bytes(out)

    @classmethod
    def FromString(cls, buf) -> "Edge":
        m = cls()
        acc: List[np.ndarray] = []
        for fno, wt, v in _iter_fields(buf):
            if fno == 1:
                m.src_node_id = int(v) & 0xFFFFFFFF
            elif fno == 2:
                m.dst_node_id = int(v) & 0xFFFFFFFF
            elif fno == 3:
                m.condensed_edge_type = int(v) & 0xFFFFFFFF
            elif fno == 4:
                _floats_from(wt, v, acc)
        m.feature_values = _cat(acc)
        return m

    def __eq__(self, o):
        return (isinstance(o, Edge) and self.src_node_id == o.src_node_id
                and self.dst_node_id == o.dst_node_id
                and self.condensed_edge_type == o.condensed_edge_type
                and np.array_equal(self.feature_values, o.feature_values))


@dataclass
class Graph:
    """graph_schema.proto:57-62 (nodes = 2, edges = 3)"""
    nodes: List[Node] = field(default_factory=list)
    edges: List[Edge] = field(default_factory=list)

    def SerializeToString(self) -> bytes:
        out = bytearray()
        for n in self.nodes:
            out += _len_delim(2, n.SerializeToString())
        for e in self.edges:
            out += _len_delim(3, e.SerializeToString())
        return bytes(out)

    @classmethod
    def FromString(cls, buf) -> "Graph":
        m = cls()
        for fno, wt, v in _iter_fields(buf):
            if fno == 2:
                m.nodes.append(Node.FromString(v))
            elif fno == 3:
                m.edges.append(Edge.FromString(v))
        return m


# ------------------------------------------------------------------------------------------------
# training_samples_schema.proto
# ------------------------------------------------------------------------------------------------


@dataclass
class Label:
    """training_samples_schema.proto:9-12"""
    label_type: str = ""
    label: int = 0

    def SerializeToString(self) -> bytes:
        out = bytearray()
        if self.label_type:
            out += _len_delim(1, self.label_type.encode("utf-8"))
        if self.label:
            out += _tag(2, 0) + _enc_varint(self.label)  # int32: negative -> 10-byte varint
        return bytes(out)

    @classmethod
    def FromString(cls, buf) -> "Label":
        m = cls()
        for fno, wt, v in _iter_fields(buf):
            if fno == 1:
                m.label_type = bytes(v).decode("utf-8")
            elif fno == 2:
                x = int(v) & 0xFFFFFFFFFFFFFFFF
                x = x - (1 << 64) if x >> 63 else x
                m.label = ((x + 2**31) % 2**32) - 2**31
        return m


@dataclass
class RootedNodeNeighborhood:
    """training_samples_schema.proto:16-19"""
    root_node: Optional[Node] = None
    neighborhood: Optional[Graph] = None

    def SerializeToString(self) -> bytes:
        out = bytearray()
        if self.root_node is not None:
            out += _len_delim(1, self.root_node.SerializeToString())
        if self.neighborhood is not None:
            out += _len_delim(2, self.neighborhood.SerializeToString())
        return bytes(out)

    @classmethod
    def FromString(cls, buf) -> "RootedNodeNeighborhood":
        m = cls()
        for fno, wt, v in _iter_fields(buf):
            if fno == 1:
                m.root_node = Node.FromString(v)
            elif fno == 2:
                m.neighborhood = Graph.FromString(v)
        return m


@dataclass
class SupervisedNodeClassificationSample:
    """training_samples_schema.proto:23-27"""
    root_node: Optional[Node] = None
    neighborhood: Optional[Graph] = None
    root_node_labels: List[Label] = field(default_factory=list)

    def SerializeToString(self) -> bytes:
        out = bytearray()
        if self.root_node is not None:
            out += _len_delim(1, self.root_node.SerializeToString())
        if self.neighborhood is not None:
            out += _len_delim(2, self.neighborhood.SerializeToString())
        for lb in self.root_node_labels:
            out += _len_delim(3, lb.SerializeToString())
        return bytes(out)

    @classmethod
    def FromString(cls, buf) -> "SupervisedNodeClassificationSample":
        m = cls()
        for fno, wt, v in _iter_fields(buf):
            if fno == 1:
                m.root_node = Node.FromString(v)
            elif fno == 2:
                m.neighborhood = Graph.FromString(v)
            elif fno == 3:
                m.root_node_labels.append(Label.FromString(v))
        return m


@dataclass
class NodeAnchorBasedLinkPredictionSample:
    """training_samples_schema.proto:31-43 (root_node=1, hard_neg_edges=2, neighborhood=3,
    pos_edges=4, neg_edges=5)"""
    root_node: Optional[Node] = None
    hard_neg_edges: List[Edge] = field(default_factory=list)
    pos_edges: List[Edge] = field(default_factory=list)
    neg_edges: List[Edge] = field(default_factory=list)
    neighborhood: Optional[Graph] = None

    def SerializeToString(self) -> bytes:
        out = bytearray()
        if self.root_node is not None:
            out += _len_delim(1, self.root_node.SerializeToString())
        for e in self.hard_neg_edges:
            out += _len_delim(2, e.SerializeToString())
        if self.neighborhood is not None:
            out += _len_delim(3, self.neighborhood.SerializeToString())
        for e in self.pos_edges:
            out += _len_delim(4, e.SerializeToString())
        for e in self.neg_edges:
            out += _len_delim(5, e.SerializeToString())
        return bytes(out)

    @classmethod
    def FromString(cls, buf) -> "NodeAnchorBasedLinkPredictionSample":
        m = cls()
        for fno, wt, v in _iter_fields(buf):
            if fno == 1:
                m.root_node = Node.FromString(v)
            elif fno == 2:
                m.hard_neg_edges.append(Edge.FromString(v))
            elif fno == 3:
                m.neighborhood = Graph.FromString(v)
            elif fno == 4:
                m.pos_edges.append(Edge.FromString(v))
            elif fno == 5:
                m.neg_edges.append(Edge.FromString(v))
        return m


# ------------------------------------------------------------------------------------------------
# tf.Example (read side only needs the three list kinds)
# ------------------------------------------------------------------------------------------------


def decode_tf_example(buf) -> Dict[str, object]:
    """tf.Example{features=1: Features{feature=1: map<string, Feature>}};
    Feature{bytes_list=1, float_list=2, int64_list=3}, each {value=1 (packed or not)}.
    Returns {name: list[bytes] | np.float32[] | np.int64[]}."""
    out: Dict[str, object] = {}
    for fno, wt, v in _iter_fields(buf):
        if fno != 1:
            continue
        for f2, _, entry in _iter_fields(v):
            if f2 != 1:
                continue
            key, feat = None, None
            for f3, _, x in _iter_fields(entry):
                if f3 == 1:
                    key = bytes(x).decode("utf-8")
                elif f3 == 2:
                    feat = x
            val: object = None
            if feat is not None:
                for kind, _, lst in _iter_fields(feat):
                    if kind == 1:
                        val = [bytes(b) for f4, _, b in _iter_fields(lst) if f4 == 1]
                    elif kind == 2:
                        acc: List[np.ndarray] = []
                        for f4, w4, b in _iter_fields(lst):
                            if f4 == 1:
                                _floats_from(w4, b, acc)
                        val = _cat(acc)
                    elif kind == 3:
                        ints: List[int] = []
                        for f4, w4, b in _iter_fields(lst):
                            if f4 != 1:
                                continue
                            if w4 == 0:
                                ints.append(int(b))
                            else:
                                mv, p = memoryview(b), 0
                                while p < len(mv):
                                    x, p = _dec_varint(mv, p)
                                    ints.append(x)
                        arr = np.array(ints, dtype=np.uint64).astype(np.int64)
                        val = arr
            out[key] = val
    return out


def encode_tf_example(features: Dict[str, object]) -> bytes:
    """inverse of decode_tf_example (keys sorted, as TensorFlow's deterministic map serialisation)"""
    entries = bytearray()
    for key in sorted(features):
        val = features[key]
        if isinstance(val, (list, tuple)) and (len(val) == 0 or isinstance(val[0], (bytes, bytearray))):
            lst = b"".join(_len_delim(1, bytes(b)) for b in val)
            feat = _len_delim(1, lst)
        else:
            arr = np.asarray(val)
            if arr.dtype.kind == "f":
                a = arr.astype("<f4")
                feat = _len_delim(2, _len_delim(1, a.tobytes()) if a.size else b"")
            else:
                payload = b"".join(_enc_varint(int(x)) for x in arr.astype(np.int64).tolist())
                feat = _len_delim(3, _len_delim(1, payload) if arr.size else b"")
        entry = _len_delim(1, key.encode("utf-8")) + _len_delim(2, feat)
        entries += _len_delim(1, entry)
    return _len_delim(1, bytes(entries))


# ------------------------------------------------------------------------------------------------
# TFRecord framing
# ------------------------------------------------------------------------------------------------


def _make_crc32c_table() -> np.ndarray:
    poly = 0x82F63B78
    tbl = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        tbl[i] = c
    return tbl


_CRC_TABLE = [int(x) for x in _make_crc32c_table()]


def crc32c(data: bytes) -> int:
    """CRC-32C (Castagnoli), table-driven."""
    c = 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in data:
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def tfrecord_frame(payload: bytes) -> bytes:
    hdr = struct.pack("<Q", len(payload))
    return hdr + struct.pack("<I", masked_crc32c(hdr)) + payload + struct.pack("<I", masked_crc32c(payload))


def read_tfrecords(path: str, verify_crc: bool = True) -> Iterator[bytes]:
    with open(path, "rb") as fh:
        data = fh.read()
    yield from iter_tfrecords(data, verify_crc)


def iter_tfrecords(data: bytes, verify_crc: bool = True) -> Iterator[bytes]:
    pos, n = 0, len(data)
    while pos < n:
        if pos + 12 > n:
            raise ValueError("truncated TFRecord header")
        hdr = data[pos:pos + 8]
        (ln,) = struct.unpack("<Q", hdr)
        (hcrc,) = struct.unpack("<I", data[pos + 8:pos + 12])
        if verify_crc and hcrc != masked_crc32c(hdr):
            raise ValueError("TFRecord length CRC mismatch")
        pos += 12
        if pos + ln + 4 > n:
            raise ValueError("truncated TFRecord payload")
        payload = data[pos:pos + ln]
        (pcrc,) = struct.unpack("<I", data[pos + ln:pos + ln + 4])
        if verify_crc and pcrc != masked_crc32c(payload):
            raise ValueError("TFRecord payload CRC mismatch")
        pos += ln + 4
        yield payload


def write_tfrecords(path: str, payloads) -> int:
    n = 0
    with open(path, "wb") as fh:
        for p in payloads:
            fh.write(tfrecord_frame(p))
            n += 1
    return n
