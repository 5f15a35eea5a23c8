"""Native reader of the Data Preprocessor's node / edge tables (tf.Example TFRecords) — host C++ inside libgigl_hip.so
(csrc/ingest.hip: gigl_tfrecord_index + gigl_tfexample_decode), multi-threaded over records.

Mirrors the reads of loadNodeDataframeIntoSparkSql / loadEdgeDataframeIntoSparkSql
(scala/subgraph_sampler/src/main/scala/libs/task/pureSpark/SGSPureSparkV1Task.scala:52-118,120-216): the id columns
are int64 lists, the feature columns named by `feature_keys` are concatenated in order into one float row
(integer-typed feature columns are cast to float), label columns are optional per record."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import COL_F32, COL_I64, GiglColumn


def read_columns(files: Sequence[str], columns: Sequence[Tuple[str, int, int]], verify_crc: bool = True,
                 n_threads: int = 0) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """columns: (feature name, COL_I64 | COL_F32, width).  -> ({name: [n, width] array}, {name: int32[n] values
    present per record}) over the records of all files, in file order."""
    lib = _lib.load()
    n_threads = n_threads or min(16, os.cpu_count() or 1)
    parts: List[Dict[str, np.ndarray]] = []
    cparts: List[Dict[str, np.ndarray]] = []
    for path in files:
        raw = np.fromfile(path, dtype=np.uint8)
        buf = C.c_void_p(raw.ctypes.data)
        n = C.c_int64()
        rc = lib.gigl_tfrecord_index(buf, raw.size, 1 if verify_crc else 0, 0, None, None, C.byref(n))
        if rc != 0:
            raise ValueError(f"{path}: truncated TFRecord file or CRC mismatch")
        off = np.empty(max(n.value, 1), dtype=np.int64)
        ln = np.empty(max(n.value, 1), dtype=np.int64)
        lib.gigl_tfrecord_index(buf, raw.size, 0, n.value, C.c_void_p(off.ctypes.data), C.c_void_p(ln.ctypes.data),
                                C.byref(n))
        out = {name: np.empty((n.value, width), dtype=np.int64 if kind == COL_I64 else np.float32)
               for name, kind, width in columns}
        cnt = {name: np.empty(n.value, dtype=np.int32) for name, _, _ in columns}
        if n.value:
            cols = (GiglColumn * len(columns))()
            for c, (name, kind, width) in zip(cols, columns):
                c.name, c.kind, c.width = name.encode("utf-8"), kind, width
                c.out, c.counts = out[name].ctypes.data, cnt[name].ctypes.data
            bad = C.c_int64(-1)
            rc = lib.gigl_tfexample_decode(buf, C.c_void_p(off.ctypes.data), C.c_void_p(ln.ctypes.data), n.value, cols,
                                           len(columns), n_threads, C.byref(bad))
            if rc != 0:
                raise ValueError(f"{path}: record {bad.value} is malformed or holds a feature of the wrong kind")
        parts.append(out)
        cparts.append(cnt)
    if not parts:
        return ({name: np.empty((0, width), dtype=np.int64 if kind == COL_I64 else np.float32)
                 for name, kind, width in columns}, {name: np.empty(0, np.int32) for name, _, _ in columns})
    return ({k: np.concatenate([p[k] for p in parts]) for k in parts[0]},
            {k: np.concatenate([p[k] for p in cparts]) for k in cparts[0]})


def feature_widths(path: str, keys: Sequence[str]) -> List[int]:
    """number of values of each feature key in the first record of `path` (the reference takes the width from the
    preprocessed schema; the tables are dense, so the first record fixes it)"""
    from . import wire
    for rec in wire.read_tfrecords(path):
        ex = wire.decode_tf_example(rec)
        return [int(np.asarray(ex[k]).size) if ex.get(k) is not None else 0 for k in keys]
    return [0 for _ in keys]
