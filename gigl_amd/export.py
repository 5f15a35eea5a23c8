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

import json
import os
import queue
import threading
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
    """`add_embedding` only ENQUEUES work: the batch is encoded on the device on the caller's stream and a writer thread
    takes it from there — waits for the encode, copies the finished blocks to the host on its own copy stream through
    two pinned staging buffers and appends them to the shard file — so the caller's next batch computes while this one
    is written.  `flush_embeddings` (and context exit) wait for everything enqueued so far and close the shard.  One
    random sync marker serves all shards of an exporter (any 16 bytes are a valid marker)."""

    def __init__(self, export_dir, file_prefix: Optional[str] = None, min_shard_size_threshold_bytes: int = 0,
                 engine=None, keep_on_device: bool = False):
        """keep_on_device: the encoded data blocks stay in HBM (`device_blocks`: (uint8 buffer, [total bytes | status])
        pairs in order) instead of being written to shard files — for a consumer on the device, and for measuring the
        path without the host leg"""
        self._keep_on_device = bool(keep_on_device)
        self.device_blocks: list = []
        if min_shard_size_threshold_bytes < 0:
            raise ValueError(
                f"file_flush_threshold must be a non-negative integer, but got {min_shard_size_threshold_bytes}")
        uri = os.fspath(getattr(export_dir, "uri", export_dir))
        if "://" in uri and not uri.startswith("file://"):
            raise ValueError(f"EmbeddingExporter writes to a local or mounted directory, got {uri!r}")
        self._export_dir = uri[len("file://"):] if uri.startswith("file://") else uri
        self._engine = engine
        self._sync_marker = os.urandom(16)
        self._num_records_written = 0
        self._num_files_written = 0
        self._in_context = False
        self._write_time = 0.0
        self._prefix = file_prefix
        self._min_shard_size_threshold_bytes = min_shard_size_threshold_bytes
        self.files_written: list = []
        self.bytes_written = 0
        # where the writer thread's time went (seconds): waiting for the device encode, device -> host, file append
        self.trace = {"wait_encode_s": 0.0, "d2h_s": 0.0, "file_s": 0.0, "batches": 0, "enqueue_s": 0.0}
        # writer thread state
        self._q: "queue.Queue" = queue.Queue(maxsize=6)  # (bounds the encoded batches waiting in HBM)
        self._thread: Optional[threading.Thread] = None
        self._err: Optional[BaseException] = None
        self._fh = None
        self._part_path = self._final_path = None
        self._shard_bytes = 0

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
        self._raise_pending()
        eng = self._eng()
        cur = torch.cuda.current_stream(eng.device)
        if eng._stream.cuda_stream != cur.cuda_stream:  # encode where the embeddings were produced
            eng.bind_stream(cur)
        out, _, scal = eng.encode_avro_embeddings_async(id_batch, embedding_batch, embedding_type, self._sync_marker)
        if self._keep_on_device:
            self.device_blocks.append((out, scal))
            self._write_time += time.perf_counter() - start
            return
        done = torch.cuda.Event()
        done.record(cur)
        if self._thread is None:
            self._thread = threading.Thread(target=self._writer, name="gigl-avro-writer", daemon=True)
            self._thread.start()
        self._q.put(("blocks", out, scal, done, int(id_batch.numel())))
        self._write_time += time.perf_counter() - start
        self.trace["enqueue_s"] += time.perf_counter() - start

    # ---- writer thread
    def _writer(self) -> None:
        dev = self._eng().device
        copy_stream = torch.cuda.Stream(device=dev)
        stage = [None, None]
        k = 0
        while True:
            item = self._q.get()
            try:
                if item[0] == "stop":
                    return
                if item[0] == "flush":
                    if self._err is None:
                        self._close_shard()
                    continue
                if self._err is not None:
                    continue  # (drain: the error is raised on the caller's side)
                _, out, scal, done, n = item
                t0 = time.perf_counter()
                done.synchronize()
                t1 = time.perf_counter()
                with torch.cuda.stream(copy_stream):
                    total, status = (int(v) for v in scal.to("cpu", non_blocking=False).tolist())
                    if status & 0xFFFFFFFF:
                        raise RuntimeError("gigl_avro_embeddings_encode: output capacity too small (status=1)")
                    buf = stage[k]
                    if buf is None or buf.numel() < total:
                        stage[k] = buf = torch.empty(max(total, 1 << 20), dtype=torch.uint8, pin_memory=True)
                    host = buf[:total]
                    host.copy_(out[:total], non_blocking=True)
                    copy_stream.synchronize()
                del out, scal
                t2 = time.perf_counter()
                self._append(memoryview(host.numpy()), n)
                t3 = time.perf_counter()
                tr = self.trace
                tr["wait_encode_s"] += t1 - t0
                tr["d2h_s"] += t2 - t1
                tr["file_s"] += t3 - t2
                tr["batches"] += 1
                k ^= 1
            except BaseException as e:  # noqa: BLE001 — surfaced by the next add / flush on the caller's thread
                self._err = e
            finally:
                if item[0] == "flush":
                    item[1].set()
                self._q.task_done()

    def _append(self, data: memoryview, n_records: int) -> None:
        if self._fh is None:
            filename = (f"shard_{self._num_files_written:08}.avro" if not self._prefix
                        else f"{self._prefix}_{self._num_files_written:08}.avro")
            os.makedirs(self._export_dir, exist_ok=True)
            self._final_path = os.path.join(self._export_dir, filename)
            self._part_path = self._final_path + ".part"
            self._fh = open(self._part_path, "wb")
            self._fh.write(avro_file_header(self._sync_marker))
            self._shard_bytes = self._fh.tell()
        self._fh.write(data)
        self._shard_bytes += len(data)
        self.bytes_written += len(data)
        self._num_records_written += n_records
        if self._min_shard_size_threshold_bytes and self._shard_bytes >= self._min_shard_size_threshold_bytes:
            self._close_shard()

    def _close_shard(self) -> None:
        if self._fh is None:
            return
        self._fh.close()
        self._fh = None
        os.replace(self._part_path, self._final_path)
        self.files_written.append(self._final_path)
        self._num_files_written += 1
        self._num_records_written = 0
        self._write_time = 0.0

    def _raise_pending(self) -> None:
        if self._err is not None:
            err, self._err = self._err, None
            raise err

    def flush_embeddings(self):
        """waits for every batch added so far and writes the buffered shard; a no-op when nothing was added since the
        last flush"""
        if self._thread is not None:
            ev = threading.Event()
            self._q.put(("flush", ev))
            ev.wait()
        self._raise_pending()

    def close(self) -> None:
        self.flush_embeddings()
        if self._thread is not None:
            self._q.put(("stop",))
            self._thread.join()
            self._thread = None

    def __del__(self):
        try:
            if self._thread is not None:
                self._q.put(("stop",))
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        if self._in_context:
            raise RuntimeError(f"{type(self).__name__} is already in a context. Do not call "
                               f"`with {type(self).__name__}:` in a nested manner.")
        self._in_context = True
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.flush_embeddings()
        self._in_context = False
