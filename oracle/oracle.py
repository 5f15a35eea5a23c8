"""ctypes/numpy front-end of oracle/gigl_oracle.c (TEST INFRASTRUCTURE ONLY — see __init__)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

INVALID = 0xFFFFFFFF
_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgigl_oracle.so")
_lib = None


def ensure_built(force: bool = False) -> str:
    src = os.path.join(_HERE, "gigl_oracle.c")
    stale = (not os.path.exists(_SO)) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)
    )
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "libgigl_oracle.so"])
    return _SO


def _L():
    global _lib
    if _lib is None:
        ensure_built()
        lib = C.CDLL(_SO)
        lib.gigl_oracle_xxh64_int32.restype = C.c_uint64
        lib.gigl_oracle_xxh64_int32.argtypes = [C.c_int32, C.c_uint64]
        _lib = lib
    return _lib


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def xxh64_int32(x: int, seed: int = 42) -> int:
    """Spark `xxhash64(int)`; returned as unsigned 64-bit python int."""
    x = ((int(x) + 2**31) % 2**32) - 2**31
    return int(_L().gigl_oracle_xxh64_int32(x, seed))


def hash_permutation(sorted_arr, internal_seed: int, sampling_seed: int = 42, counter: int = 1):
    a = np.ascontiguousarray(sorted_arr, dtype=np.uint32)
    out = np.empty_like(a)
    iseed = ((int(internal_seed) + 2**31) % 2**32) - 2**31
    rc = _L().gigl_oracle_hash_permutation(
        _p(a, C.c_uint32), C.c_int64(a.size), C.c_int32(iseed), C.c_int32(sampling_seed),
        C.c_int32(counter), _p(out, C.c_uint32))
    assert rc == 0
    return out


def sample_khop(rowptr, col, roots, fanouts: Sequence[int], sampling_seed: int = 42,
                first_counter: int = 1, canonical: bool = False) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """Tree-layout k-hop sample; per-parent order = permutation order (reference slice order) or, with
    canonical=True, ascending ids (the HIP path's form).  Returns (nbr[k], cnt[k]) lists."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.uint32)
    roots = np.ascontiguousarray(roots, dtype=np.uint32)
    b, hops = roots.size, len(fanouts)
    f = np.asarray(fanouts, dtype=np.int32)
    nbr, cnt, parents = [], [], b
    for k in range(hops):
        cnt.append(np.zeros(max(parents, 0), dtype=np.int32))
        parents *= int(f[k])
        nbr.append(np.full(max(parents, 0), INVALID, dtype=np.uint32))
    nbr_p = (C.POINTER(C.c_uint32) * hops)(*[_p(x, C.c_uint32) for x in nbr])
    cnt_p = (C.POINTER(C.c_int32) * hops)(*[_p(x, C.c_int32) for x in cnt])
    rc = _L().gigl_oracle_sample_khop(
        C.c_int64(rowptr.size - 1), _p(rowptr, C.c_int64), _p(col, C.c_uint32),
        _p(roots, C.c_uint32), C.c_int32(b), _p(f, C.c_int32), C.c_int32(hops),
        C.c_int32(sampling_seed), C.c_int32(first_counter), C.c_int32(1 if canonical else 0), nbr_p, cnt_p)
    assert rc == 0, rc
    return nbr, cnt


def tree_edges(roots, fanouts: Sequence[int], nbr: Sequence[np.ndarray]):
    """per-root set of (src, dst) global-id edges of a tree-layout sample (order-independent view)"""
    roots = np.asarray(roots, dtype=np.uint32)
    out = [set() for _ in range(roots.size)]
    per_root = 1
    parent = None
    for k, f in enumerate(fanouts):
        per_root *= int(f)
        a = nbr[k]
        for j in np.nonzero(a != INVALID)[0]:
            p = j // int(f)
            dst = roots[p] if k == 0 else parent[p]
            out[j // per_root].add((int(a[j]), int(dst)))
        parent = a
    return out


def collate_reference(node_lists: Sequence[np.ndarray], edge_lists: Sequence[Tuple[np.ndarray, np.ndarray]],
                      coalesce: bool = True):
    """Reference collate (first-seen remap, edge dedup, coalesce).  Returns (nodes, src_local, dst_local)."""
    n = len(node_lists)
    node_off = np.zeros(n + 1, dtype=np.int64)
    edge_off = np.zeros(n + 1, dtype=np.int64)
    for i in range(n):
        node_off[i + 1] = node_off[i] + len(node_lists[i])
        edge_off[i + 1] = edge_off[i] + len(edge_lists[i][0])
    cat = lambda xs: (np.concatenate([np.asarray(x, dtype=np.uint32) for x in xs])
                      if len(xs) else np.zeros(0, np.uint32)).astype(np.uint32)
    nodes = np.ascontiguousarray(cat(node_lists))
    es = np.ascontiguousarray(cat([e[0] for e in edge_lists]))
    ed = np.ascontiguousarray(cat([e[1] for e in edge_lists]))
    out_nodes = np.zeros(max(nodes.size, 1), dtype=np.uint32)
    out_s = np.zeros(max(es.size, 1), dtype=np.int64)
    out_d = np.zeros(max(es.size, 1), dtype=np.int64)
    nn, ne = C.c_int64(0), C.c_int64(0)
    rc = _L().gigl_oracle_collate_reference(
        C.c_int32(n), _p(node_off, C.c_int64), _p(nodes, C.c_uint32), _p(edge_off, C.c_int64),
        _p(es, C.c_uint32), _p(ed, C.c_uint32), C.c_int32(1 if coalesce else 0),
        _p(out_nodes, C.c_uint32), C.byref(nn), _p(out_s, C.c_int64), _p(out_d, C.c_int64),
        C.byref(ne))
    if rc == -1:
        raise TypeError("Tried to fetch a node which we have no information on")
    assert rc == 0, rc
    return out_nodes[: nn.value].copy(), out_s[: ne.value].copy(), out_d[: ne.value].copy()


def union_build(roots, fanouts: Sequence[int], nbr: Sequence[np.ndarray]):
    """Level-ordered union graph (the numbering gigl_union_build produces).
    Returns dict(meta, nodes, rowptr, col, root_local)."""
    roots = np.ascontiguousarray(roots, dtype=np.uint32)
    b, hops = roots.size, len(fanouts)
    f = np.asarray(fanouts, dtype=np.int32)
    nbr = [np.ascontiguousarray(x, dtype=np.uint32) for x in nbr]
    total = b + sum(x.size for x in nbr)
    meta = np.zeros(16, dtype=np.int32)
    nodes = np.zeros(max(total, 1), dtype=np.uint32)
    rowptr = np.zeros(total + 2, dtype=np.int32)
    col = np.zeros(max(total, 1), dtype=np.int32)
    root_local = np.zeros(max(b, 1), dtype=np.int32)
    nbr_p = (C.POINTER(C.c_uint32) * hops)(*[_p(x, C.c_uint32) for x in nbr])
    rc = _L().gigl_oracle_union_build(
        _p(roots, C.c_uint32), C.c_int32(b), _p(f, C.c_int32), C.c_int32(hops), nbr_p,
        _p(meta, C.c_int32), _p(nodes, C.c_uint32), _p(rowptr, C.c_int32), _p(col, C.c_int32),
        _p(root_local, C.c_int32))
    assert rc == 0, rc
    nn, ne = int(meta[0]), int(meta[1])
    return dict(meta=meta, nodes=nodes[:nn].copy(), rowptr=rowptr[: nn + 1].copy(),
                col=col[:ne].copy(), root_local=root_local[:b].copy())


def build_csc(n: int, src, dst, is_directed: bool, keep_multi_edges: bool = False):
    """keep_multi_edges (directed only): repeated (src, dst) records stay — rows are ascending multisets"""
    src = np.ascontiguousarray(src, dtype=np.uint32)
    dst = np.ascontiguousarray(dst, dtype=np.uint32)
    e_out = C.c_int64(0)
    args = [C.c_int64(n), C.c_int64(src.size), _p(src, C.c_uint32), _p(dst, C.c_uint32),
            C.c_int32((2 if keep_multi_edges else 1) if is_directed else 0)]
    rc = _L().gigl_oracle_build_csc(*args, None, None, C.byref(e_out))
    assert rc == 0, rc
    rowptr = np.zeros(n + 1, dtype=np.int64)
    col = np.zeros(max(e_out.value, 1), dtype=np.uint32)
    rc = _L().gigl_oracle_build_csc(*args, _p(rowptr, C.c_int64), _p(col, C.c_uint32), C.byref(e_out))
    assert rc == 0, rc
    return rowptr, col[: e_out.value].copy()


def murmur3_x86_32(data: bytes, seed: int) -> int:
    """MurmurHash3_x86_32 (signed int32)"""
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    lib = _L()
    lib.gigl_oracle_murmur3_x86_32.restype = C.c_int32
    return int(lib.gigl_oracle_murmur3_x86_32(_p(np.ascontiguousarray(buf), C.c_uint8), C.c_int64(len(data)),
                                              C.c_uint32(seed & 0xFFFFFFFF)))


def split_slots(a, b=None, condensed_type: int = 0, symmetric: bool = False) -> np.ndarray:
    """hash slots in [0, 10000) of node keys "<a>-<type>" (b None) or edge keys "<a>-<type>-<b>"
    (HashingAssigner, AbstractAssigners.scala:30-111)"""
    a = np.ascontiguousarray(a, dtype=np.uint32)
    out = np.zeros(a.size, dtype=np.int32)
    if b is None:
        rc = _L().gigl_oracle_split_slots(_p(a, C.c_uint32), None, C.c_int64(a.size), C.c_int32(condensed_type),
                                          C.c_int32(0), C.c_int32(0), _p(out, C.c_int32))
    else:
        bb = np.ascontiguousarray(b, dtype=np.uint32)
        rc = _L().gigl_oracle_split_slots(_p(a, C.c_uint32), _p(bb, C.c_uint32), C.c_int64(a.size),
                                          C.c_int32(condensed_type), C.c_int32(1), C.c_int32(1 if symmetric else 0),
                                          _p(out, C.c_int32))
    assert rc == 0
    return out
