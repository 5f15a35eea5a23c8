"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the Avro encoding the reference's
EmbeddingExporter produces through fastavro (python/gigl/common/data/export.py:34-43 schema, :103-135 records).

fastavro is a third-party dependency of the reference (python/pyproject.toml) and is not installed in this image, so
the restatement follows the published Apache Avro 1.x specification ("Binary Encoding", "Object Container Files") and
is pinned on the specification's own examples (tests/test_avro_oracle.py: zig-zag table, "foo", the [3, 27] array) and
on the decoded records the reference's unit test expects (python/tests/unit/common/data/export_test.py:63-111).
Byte-level parity with fastavro's output is therefore UNPINNED (block boundaries, metadata order and the random sync
marker are writer choices every conforming reader ignores); record-level parity is what the tests assert."""
import json
import struct

import numpy as np

MAGIC = b"Obj\x01"


def zigzag(v: int) -> int:
    return ((v << 1) ^ (v >> 63)) & 0xFFFFFFFFFFFFFFFF


def encode_long(v: int) -> bytes:
    z = zigzag(int(v))
    out = bytearray()
    while z >= 0x80:
        out.append((z & 0x7F) | 0x80)
        z >>= 7
    out.append(z)
    return bytes(out)


def encode_string(s: str) -> bytes:
    b = s.encode("utf-8")
    return encode_long(len(b)) + b


def encode_array(items, item_encoder) -> bytes:
    out = bytearray()
    if len(items) > 0:
        out += encode_long(len(items))
        for it in items:
            out += item_encoder(it)
    out += encode_long(0)
    return bytes(out)


def encode_float(x) -> bytes:
    return struct.pack("<f", float(x))


def encode_embedding_record(node_id: int, node_type: str, emb) -> bytes:
    return encode_long(node_id) + encode_string(node_type) + encode_array(list(emb), encode_float)


def encode_block(records, sync: bytes) -> bytes:
    body = b"".join(records)
    return encode_long(len(records)) + encode_long(len(body)) + body + sync


def encode_embedding_blocks(ids, emb, node_type: str, sync: bytes, records_per_block: int) -> bytes:
    """the data blocks of one add_embedding call with a fixed number of records per block"""
    recs = [encode_embedding_record(int(i), node_type, np.asarray(e, dtype=np.float32)) for i, e in zip(ids, emb)]
    out = bytearray()
    for s in range(0, len(recs), records_per_block):
        out += encode_block(recs[s:s + records_per_block], sync)
    return bytes(out)


class _Reader:
    def __init__(self, data: bytes):
        self.d, self.p = data, 0

    def long(self) -> int:
        z, shift = 0, 0
        while True:
            b = self.d[self.p]
            self.p += 1
            z |= (b & 0x7F) << shift
            if not b & 0x80:
                break
            shift += 7
        return (z >> 1) ^ -(z & 1)

    def bytes_(self) -> bytes:
        n = self.long()
        v = self.d[self.p:self.p + n]
        assert len(v) == n, "truncated"
        self.p += n
        return v

    def eof(self) -> bool:
        return self.p >= len(self.d)


def read_embedding_file(data: bytes):
    """object container file -> (schema dict, [{node_id, node_type, emb (list of float)} ...]); every block's byte size
    and sync marker is checked"""
    r = _Reader(data)
    assert data[:4] == MAGIC, "not an Avro object container file"
    r.p = 4
    meta = {}
    while True:
        n = r.long()
        if n == 0:
            break
        if n < 0:
            n = -n
            r.long()
        for _ in range(n):
            k = r.bytes_().decode()
            meta[k] = r.bytes_()
    assert meta.get("avro.codec", b"null") == b"null"
    schema = json.loads(meta["avro.schema"])
    sync = data[r.p:r.p + 16]
    r.p += 16
    out = []
    while not r.eof():
        count = r.long()
        size = r.long()
        end = r.p + size
        for _ in range(count):
            node_id = r.long()
            node_type = r.bytes_().decode()
            emb = []
            while True:
                n = r.long()
                if n == 0:
                    break
                if n < 0:
                    n = -n
                    r.long()
                emb.extend(np.frombuffer(data, dtype="<f4", count=n, offset=r.p).tolist())
                r.p += 4 * n
            out.append({"node_id": node_id, "node_type": node_type, "emb": emb})
        assert r.p == end, "block byte size does not match its records"
        assert data[r.p:r.p + 16] == sync, "sync marker mismatch"
        r.p += 16
    return schema, out
