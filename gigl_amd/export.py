"""EmbeddingExporter — the reference's inference output writer (python/gigl/common/data/export.py:52-209) with the
record encoding done on the device.

Same interface and behaviour as the reference class: `add_embedding(id_batch, embedding_batch, embedding_type)`
appends to an in-memory shard buffer, the buffer becomes `shard_%08d.avro` (or `<prefix>_%08d.avro`) on
`flush_embeddings()` / context exit / once it holds `min_shard_size_threshold_bytes`; nested `with` raises; a
negative threshold raises the reference's message.  The reference builds one Python dict per record and hands them
to fastavro; here `gigl_avro_embeddings_encode` (csrc/export.hip) turns the id and embedding tensors — usually
still in HBM, straight out of the model — into finished Avro data blocks and the host only prepends the file header.
Like fastavro appending to a non-empty buffer, every `add_embedding` after the first re-uses the shard's header and
sync marker and appends data blocks.

Shards are written to a local / mounted directory: object stores are outside this build's scope, so a
`gs://` export_dir raises instead of silently writing somewhere else.  `load_embeddings_to_bigquery` (export.py:212-259)
is control-plane and not provided."""
from __future__ import annotations

import io
import json
import os
import time
from typing import Optional

import torch

_NODE_ID_KEY = "node_id"
_EMBEDDING_TYPE_KEY = "node_type"
_EMBEDDING_KEY = "emb"

# python/gigl/common/data/export.py:34-43
AVRO_SCHEMA = {
    "type": "record",
    "name": "Embedding",
    "fields": [
        {"name": _NODE_ID_KEY, "type": "long"},
        {"name": _EMBEDDING_TYPE_KEY, "type": "string"},
        {"name": _EMBEDDING_KEY, "type": {"type": "array", "items": "float"}},
    ],
}

_MAGIC = b"Obj\x01"


def _avro_long(v: int) -> bytes:
    z = ((v << 1) ^ (v >> 63)) & 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while z >= 0x80:
        out.append((z & 0x7F) | 0x80)
        z >>= 7
    out.append(z)
    return bytes(out)


def avro_file_header(sync_marker: bytes, schema=AVRO_SCHEMA, codec: str = "null") -> bytes:
    """object container file header: magic, metadata map {avro.schema, avro.codec}, 16-byte sync marker"""
    assert len(sync_marker) == 16
    meta = {"avro.schema": json.dumps(schema).encode(), "avro.codec": codec.encode()}
    out = bytearray(_MAGIC)
    out += _avro_long(len(meta))
    for k, v in meta.items():
        kb = k.encode()
        out += _avro_long(len(kb)) + kb + _avro_long(len(v)) + v
    out += _avro_long(0)
    out += sync_marker
    return bytes(out)


class EmbeddingExporter:
    def __init__(self, export_dir, file_prefix: Optional[str] = None, min_shard_size_threshold_bytes: int = 0,
                 engine=None):
        if min_shard_size_threshold_bytes < 0:
            raise ValueError(
                f"file_flush_threshold must be a non-negative integer, but got {min_shard_size_threshold_bytes}")
        uri = os.fspath(getattr(export_dir, "uri", export_dir))
        if "://" in uri and not uri.startswith("file://"):
            raise ValueError(f"EmbeddingExporter writes to a local or mounted directory, got {uri!r}")
        self._export_dir = uri[len("file://"):] if uri.startswith("file://") else uri
        self._engine = engine
        self._buffer = io.BytesIO()
        self._sync_marker: Optional[bytes] = None
        self._num_records_written = 0
        self._num_files_written = 0
        self._in_context = False
        self._write_time = 0.0
        self._prefix = file_prefix
        self._min_shard_size_threshold_bytes = min_shard_size_threshold_bytes
        self.files_written: list = []

    def _eng(self):
        if self._engine is None:
            from .engine import HipEngine  # raises when libgigl_hip.so or the GPU is missing: there is no host encoder
            self._engine = HipEngine(torch.cuda.current_device() if torch.cuda.is_available() else 0)
        return self._engine

    def add_embedding(self, id_batch: torch.Tensor, embedding_batch: torch.Tensor, embedding_type: str):
        """ids [n] (any integer dtype) and embeddings [n, D] (any real dtype; written as Avro `float`), host or device"""
        start = time.perf_counter()
        if id_batch.dim() != 1 or embedding_batch.dim() != 2 or embedding_batch.shape[0] != id_batch.numel():
            raise ValueError(f"expected ids [n] and embeddings [n, D], got {tuple(id_batch.shape)} and "
                             f"{tuple(embedding_batch.shape)}")
        if id_batch.is_floating_point():
            raise TypeError("node ids must be integers")
        eng = self._eng()
        if self._buffer.tell() == 0:
            self._sync_marker = os.urandom(16)
            self._buffer.write(avro_file_header(self._sync_marker))
        blocks, _ = eng.encode_avro_embeddings(id_batch, embedding_batch, embedding_type, self._sync_marker)
        self._buffer.write(memoryview(self._to_host(blocks).numpy()))
        self._num_records_written += int(id_batch.numel())
        self._write_time += time.perf_counter() - start
        if self._min_shard_size_threshold_bytes and self._buffer.tell() >= self._min_shard_size_threshold_bytes:
            self.flush_embeddings()

    def _to_host(self, blocks: torch.Tensor) -> torch.Tensor:
        """device -> host through a reusable pinned staging buffer (pageable copies run at a fraction of the link)"""
        n = int(blocks.numel())
        stage = getattr(self, "_stage", None)
        if stage is None or stage.numel() < n:
            self._stage = stage = torch.empty(max(n, 1 << 20), dtype=torch.uint8, pin_memory=True)
        out = stage[:n]
        out.copy_(blocks, non_blocking=True)
        torch.cuda.current_stream(blocks.device).synchronize()
        return out

    def _flush(self):
        filename = (f"shard_{self._num_files_written:08}.avro" if not self._prefix
                    else f"{self._prefix}_{self._num_files_written:08}.avro")
        os.makedirs(self._export_dir, exist_ok=True)
        path = os.path.join(self._export_dir, filename)
        tmp = path + ".part"
        with open(tmp, "wb") as f:
            f.write(self._buffer.getbuffer())
        os.replace(tmp, path)
        self.files_written.append(path)
        self._num_files_written += 1
        self._buffer = io.BytesIO()
        self._num_records_written = 0
        self._write_time = 0.0

    def flush_embeddings(self):
        """writes the buffered shard; a no-op when nothing was added since the last flush"""
        if self._buffer.tell() == 0:
            return
        self._flush()

    def __enter__(self):
        if self._in_context:
            raise RuntimeError(f"{type(self).__name__} is already in a context. Do not call "
                               f"`with {type(self).__name__}:` in a nested manner.")
        self._in_context = True
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.flush_embeddings()
        self._in_context = False
