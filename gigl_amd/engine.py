"""HipEngine — thin host-side owner of one libgigl_hip ctx (one device, one stream).

torch is used only for device memory and streams (`torch.empty(..., device=...)`, `.data_ptr()`,
the current stream handle); all compute happens in the HIP library.  Lifecycle mirrors the
reference's per-partition `KHopSamplerService.setup()/teardown()`
(scala_spark35/common/src/main/scala/graphdb/KHopSamplerService.scala:10-33).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import (DTYPE_F16, DTYPE_F32, GIGL_INVALID, GIGL_META_LEN, GiglTree, GiglUnion, LOC_DEVICE,
                   LOC_HOST, MODE_FAST, MODE_SPARK_HASH, check)


@dataclass
class Tree:
    """device-resident tree-layout sample (include/gigl_hip.h `gigl_tree`)"""
    roots: torch.Tensor            # int32 storage of uint32 ids, [b]
    fanouts: List[int]
    nbr: List[torch.Tensor]        # int32 storage of uint32 ids, [slots_k]; -1 == GIGL_INVALID
    cnt: List[torch.Tensor]        # int32 [parents_k]
    c_struct: GiglTree

    @property
    def b(self) -> int:
        return int(self.roots.numel())

    def sampled_edges(self) -> int:
        return int(sum(int(c.sum().item()) for c in self.cnt))


@dataclass
class UnionGraph:
    """device-resident batch union graph (include/gigl_hip.h `gigl_union`)"""
    meta: torch.Tensor        # int32 [16]
    nodes: torch.Tensor       # int32 storage of uint32 global ids [cap_nodes]
    rowptr: torch.Tensor      # int32 [cap_nodes+1]  row i = col[rowptr[i]:rowend[i]]
    rowend: torch.Tensor      # int32 [cap_nodes+1]
    col: torch.Tensor         # int32 [cap_edges]
    root_local: torch.Tensor  # int32 [b]
    hops: int
    c_struct: GiglUnion

    def counts(self):
        m = self.meta.cpu().tolist()
        if m[8]:
            raise RuntimeError(f"{m[8]} union rows exceeded the in-LDS dedup capacity (meta[GIGL_META_OVERFLOW])")
        return dict(n_nodes=m[0], n_edges=m[1], levels=m[2:2 + self.hops + 1])

    def to_csr(self):
        """host copy as a packed CSR: (nodes uint32[n], rowptr int64[n+1], col int32[e])"""
        m = self.counts()
        n = m["n_nodes"]
        rp = self.rowptr[: n + 1].cpu().numpy().astype(np.int64)
        re = self.rowend[: n + 1].cpu().numpy().astype(np.int64)
        col = self.col.cpu().numpy()
        lens = re[:n] - rp[:n]
        out_rp = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=out_rp[1:])
        idx = np.repeat(rp[:n] - out_rp[:n], lens) + np.arange(int(out_rp[-1]))
        return self.nodes[:n].cpu().numpy().view(np.uint32), out_rp, col[idx]


_DEFAULT_ENGINES = {}


def default_engine(device) -> "HipEngine":
    """one shared ctx per device for the stateless ops (loss, decoder) that the reference calls as free functions;
    bound to torch's current stream of that device at every use"""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(f"gigl_amd runs on a HIP device only (got {dev}); there is no CPU fallback")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    eng = _DEFAULT_ENGINES.get(idx)
    if eng is None or not eng._ctx:
        eng = _DEFAULT_ENGINES[idx] = HipEngine(idx)
    cur = torch.cuda.current_stream(eng.device)
    if eng._stream.cuda_stream != cur.cuda_stream:
        eng.bind_stream(cur)
    return eng


class HipEngine:
    def __init__(self, device: int = 0):
        self._lib = _lib.load()  # raises if the HIP library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("gigl_amd.HipEngine needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.device = torch.device("cuda", device)
        ctx = C.c_void_p()
        check(self._lib.gigl_ctx_create(device, C.byref(ctx)))
        self._ctx = ctx
        self._graph = None
        self._graph_out = None
        self._feat = None
        self.n_nodes = 0
        self.n_edges = 0
        self.feat_dim = 0
        self.feat_dtype = DTYPE_F32
        self.bind_stream()

    # ---- lifecycle -------------------------------------------------------------------------
    def bind_stream(self, stream: Optional[torch.cuda.Stream] = None) -> None:
        """run library kernels on `stream` (default: torch's current stream on this device)"""
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        old = getattr(self, "_stream", None)
        self._stream = s
        if old is not None and old.cuda_stream == s.cuda_stream:
            return  # (re-binding drains the old stream: skip when nothing changes)
        check(self._lib.gigl_ctx_set_stream(self._ctx, C.c_void_p(s.cuda_stream)), self._ctx)

    def synchronize(self) -> None:
        check(self._lib.gigl_ctx_synchronize(self._ctx), self._ctx)

    def profile_enable(self, kernels: Sequence[str] = (), capacity: int = 0) -> None:
        """HIP-event timing of the named kernels (see _lib.KERNEL_IDS); empty = off"""
        mask = 0
        for k in kernels:
            mask |= 1 << _lib.KERNEL_IDS[k]
        check(self._lib.gigl_profile_enable(self._ctx, mask, capacity), self._ctx)

    def profile_read(self, kernel: str):
        """-> (total_ms, launches) since the last enable/reset (synchronises the stream)"""
        ms, n = C.c_double(), C.c_int64()
        check(self._lib.gigl_profile_read(self._ctx, _lib.KERNEL_IDS[kernel], C.byref(ms), C.byref(n)), self._ctx)
        return ms.value, n.value

    def profile_reset(self) -> None:
        check(self._lib.gigl_profile_reset(self._ctx), self._ctx)

    def share_resident(self, other: "HipEngine") -> None:
        """borrow `other`'s HBM-resident graph / feature table (same device).  Lets several ctxs — one per
        stream / host thread — sample the same graph concurrently; `other` must outlive this engine."""
        assert other.device == self.device
        self._graph, self._graph_out, self._feat = other._graph, other._graph_out, other._feat
        self._feat_ptr = getattr(other, "_feat_ptr", None)
        self._efeat = getattr(other, "_efeat", None)
        self._efeat_handle = getattr(other, "_efeat_handle", None)
        self.n_nodes, self.n_edges = other.n_nodes, other.n_edges
        self.feat_dim, self.feat_dtype = other.feat_dim, other.feat_dtype
        self._borrowed = True

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            for p in list(getattr(self, "_plans", [])):
                p.close()
            for entry in getattr(self, "_label_edges", {}).values():
                self._free_label_edges(entry)
            self._label_edges = {}
            if not getattr(self, "_borrowed", False):
                for h, fn in ((self._graph, self._lib.gigl_graph_destroy),
                              (self._graph_out, self._lib.gigl_graph_destroy),
                              (self._feat, self._lib.gigl_features_destroy),
                              (getattr(self, "_efeat_handle", None), self._lib.gigl_features_destroy)):
                    if h:
                        fn(h)
            self._graph = self._graph_out = self._feat = self._efeat_handle = self._efeat = None
            self._lib.gigl_ctx_destroy(self._ctx)
            self._ctx = None

    teardown = close

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- resident data ---------------------------------------------------------------------
    @staticmethod
    def _ptr_loc(x):
        if isinstance(x, torch.Tensor):
            x = x.contiguous()
            return x, C.c_void_p(x.data_ptr()), (LOC_DEVICE if x.is_cuda else LOC_HOST)
        x = np.ascontiguousarray(x)
        return x, C.c_void_p(x.ctypes.data), LOC_HOST

    def load_csc(self, rowptr, col, *, out_graph: bool = False) -> None:
        """rowptr int64[n+1], col uint32/int32[e] (numpy or torch, host or device)."""
        rp, rp_p, loc1 = self._ptr_loc(rowptr if isinstance(rowptr, torch.Tensor) else np.asarray(rowptr, dtype=np.int64))
        if isinstance(col, torch.Tensor):
            assert col.dtype in (torch.int32, torch.uint32)
        else:
            col = np.asarray(col).astype(np.uint32, copy=False)
        cl, cl_p, loc2 = self._ptr_loc(col)
        assert loc1 == loc2, "rowptr and col must live on the same side"
        if isinstance(rp, torch.Tensor):
            assert rp.dtype == torch.int64
        n = int(rp.shape[0]) - 1
        e = int(cl.shape[0])
        g = C.c_void_p()
        check(self._lib.gigl_graph_load_csc(self._ctx, n, e, rp_p, cl_p, loc1, C.byref(g)), self._ctx)
        self._set_graph(g, n, e, out_graph)

    def build_from_coo(self, n: int, src, dst, is_directed: bool, *, out_graph: bool = False,
                       keep_multi_edges: bool = False) -> None:
        """keep_multi_edges (directed graphs): repeated (src, dst) records stay, as in the reference's directed path;
        the sampler then draws over the multiset (GIGL_DIRECTED_MULTI, include/gigl_hip.h)"""
        s, s_p, loc1 = self._ptr_loc(src if isinstance(src, torch.Tensor) else np.asarray(src).astype(np.uint32))
        d, d_p, loc2 = self._ptr_loc(dst if isinstance(dst, torch.Tensor) else np.asarray(dst).astype(np.uint32))
        assert loc1 == loc2
        g = C.c_void_p()
        mode = (2 if keep_multi_edges else 1) if is_directed else 0
        check(self._lib.gigl_graph_build_from_coo(self._ctx, n, int(s.shape[0]), s_p, d_p, loc1, mode, C.byref(g)),
              self._ctx)
        nn, ee = C.c_int64(), C.c_int64()
        check(self._lib.gigl_graph_info(g, C.byref(nn), C.byref(ee)), self._ctx)
        self._set_graph(g, nn.value, ee.value, out_graph)

    def build_shard_from_coo(self, n: int, rank: int, world: int, src, dst, is_directed: bool,
                             keep_multi_edges: bool = False) -> None:
        """this rank's shard of the graph hash-partitioned over `world` ranks, from the WHOLE edge list
        (gigl_graph_build_shard_from_coo): rows = the nodes with id % world == rank (row id // world), ids inside the
        rows stay global.  n_nodes stays the GLOBAL node count; n_edges = the shard's."""
        s, s_p, loc1 = self._ptr_loc(src if isinstance(src, torch.Tensor) else np.asarray(src).astype(np.uint32))
        d, d_p, loc2 = self._ptr_loc(dst if isinstance(dst, torch.Tensor) else np.asarray(dst).astype(np.uint32))
        assert loc1 == loc2
        g = C.c_void_p()
        mode = (2 if keep_multi_edges else 1) if is_directed else 0
        check(self._lib.gigl_graph_build_shard_from_coo(self._ctx, n, rank, world, int(s.shape[0]), s_p, d_p, loc1, mode,
                                                        C.byref(g)), self._ctx)
        nn, ee = C.c_int64(), C.c_int64()
        check(self._lib.gigl_graph_info(g, C.byref(nn), C.byref(ee)), self._ctx)
        self._set_graph(g, int(n), ee.value, False)
        self.n_shard_rows = nn.value

    def _set_graph(self, g, n, e, out_graph):
        attr = "_graph_out" if out_graph else "_graph"
        old = getattr(self, attr)
        if old:
            self._lib.gigl_graph_destroy(old)
        setattr(self, attr, g)
        if not out_graph:
            self.n_nodes, self.n_edges = n, e

    def graph_to_host(self, out_graph: bool = False):
        g = self._graph_out if out_graph else self._graph
        n, e = C.c_int64(), C.c_int64()
        check(self._lib.gigl_graph_info(g, C.byref(n), C.byref(e)), self._ctx)
        rp, cl = C.c_void_p(), C.c_void_p()
        check(self._lib.gigl_graph_device_ptrs(g, C.byref(rp), C.byref(cl)), self._ctx)
        rowptr = np.empty(n.value + 1, dtype=np.int64)
        col = np.empty(max(e.value, 1), dtype=np.uint32)
        check(self._lib.gigl_memcpy(self._ctx, C.c_void_p(rowptr.ctypes.data), LOC_HOST, rp, LOC_DEVICE,
                                    rowptr.nbytes), self._ctx)
        if e.value:
            check(self._lib.gigl_memcpy(self._ctx, C.c_void_p(col.ctypes.data), LOC_HOST, cl, LOC_DEVICE,
                                        e.value * 4), self._ctx)
        return rowptr, col[: e.value]

    # ---- edge features (S1 `_edge_features`, S6 hydrateEdges): one fp32 row per resident edge, in `col` order
    @property
    def edge_feat_dim(self) -> int:
        ef = getattr(self, "_efeat", None)
        return 0 if ef is None else int(ef.shape[1])

    def edge_ids(self, src: torch.Tensor, dst: torch.Tensor, *, out_graph: bool = False) -> torch.Tensor:
        """position of every edge src[i] -> dst[i] in the resident CSC `col` array (-1: no such edge)"""
        g = self._graph_out if out_graph else self._graph
        src = src.to(device=self.device).contiguous()
        dst = dst.to(device=self.device).contiguous()
        assert src.dtype in (torch.int32, torch.uint32) and dst.dtype == src.dtype and src.shape == dst.shape
        eid = torch.empty(src.numel(), dtype=torch.int64, device=self.device)
        check(self._lib.gigl_edge_ids(self._ctx, g, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()),
                                      src.numel(), C.c_void_p(eid.data_ptr())), self._ctx)
        return eid

    def load_edge_features(self, src, dst, feats, is_directed: bool) -> None:
        """edge feature rows given in the order of the COO list the graph was built from -> a table in `col` order.
        Undirected graphs: the row of (a,b) also serves (b,a); where several input rows name the same edge the
        first one in input order wins (the reference keeps an arbitrary one: dropDuplicates after canonicalising,
        SGSPureSparkV1Task.scala:218-258)."""
        def as_ids(v):
            v = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v).astype(np.int64))
            return v.to(self.device).to(torch.int32)
        s, d = as_ids(src), as_ids(dst)
        f = feats if isinstance(feats, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(feats))
        f = f.to(device=self.device, dtype=torch.float32)
        assert f.dim() == 2 and f.shape[0] == s.numel()
        m = s.numel()
        row = torch.arange(m, device=self.device)
        eid = self.edge_ids(s, d)
        if not is_directed:
            eid = torch.cat([eid, self.edge_ids(d, s)])
            row = torch.cat([row, row])
        self._stream.synchronize()
        keep = eid >= 0
        win = torch.full((self.n_edges,), m, dtype=torch.int64, device=self.device)
        win.scatter_reduce_(0, eid[keep], row[keep], reduce="amin")
        if bool((win == m).any()):
            raise ValueError("edge features do not cover every resident edge (was the graph built from this list?)")
        self._set_edge_table(f[win].contiguous())

    def _set_edge_table(self, table: torch.Tensor) -> None:
        """`table`: [n_edges, De] fp32 on the device, row p = features of the edge at col[p]"""
        assert table.shape[0] == self.n_edges and table.dtype == torch.float32 and table.is_cuda
        self._efeat = table
        h = C.c_void_p()
        check(self._lib.gigl_features_load(self._ctx, table.shape[0], table.shape[1], DTYPE_F32,
                                           C.c_void_p(table.data_ptr()), LOC_DEVICE, C.byref(h)), self._ctx)
        if getattr(self, "_efeat_handle", None) and not getattr(self, "_borrowed", False):
            self._lib.gigl_features_destroy(self._efeat_handle)
        self._efeat_handle = h

    def union_edge_ids(self, u: "UnionGraph") -> torch.Tensor:
        """[cap_edges] int64: resident edge id of the union edge stored at each position of u.col (-1: unused)"""
        eid = torch.empty(int(u.col.numel()), dtype=torch.int64, device=self.device)
        check(self._lib.gigl_union_edge_ids(self._ctx, self._graph, C.byref(u.c_struct), C.c_void_p(eid.data_ptr())),
              self._ctx)
        return eid

    def union_edge_attr(self, u: "UnionGraph") -> torch.Tensor:
        """[cap_edges, De] fp32 edge feature rows aligned with u.col (zeros at unused positions)"""
        if getattr(self, "_efeat", None) is None:
            raise RuntimeError("no edge features loaded (load_edge_features)")
        eid = self.union_edge_ids(u)
        with torch.cuda.stream(self._stream):
            rows = self._efeat[eid.clamp(min=0)]
            rows.masked_fill_((eid < 0).unsqueeze(1), 0.0)
        return rows

    def load_features(self, x) -> None:
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x))
        assert x.dim() == 2 and x.dtype in (torch.float32, torch.float16)
        x = x.contiguous()
        dt = DTYPE_F32 if x.dtype == torch.float32 else DTYPE_F16
        f = C.c_void_p()
        check(self._lib.gigl_features_load(self._ctx, x.shape[0], x.shape[1], dt, C.c_void_p(x.data_ptr()),
                                           LOC_DEVICE if x.is_cuda else LOC_HOST, C.byref(f)), self._ctx)
        if self._feat:
            self._lib.gigl_features_destroy(self._feat)
        self._feat = f
        self.feat_dim, self.feat_dtype = int(x.shape[1]), dt
        rows = C.c_void_p()
        check(self._lib.gigl_features_device_ptr(f, C.byref(rows), None, None, None), self._ctx)
        self._feat_ptr = rows

    # ---- per-batch ops (device pointers only, no host sync) --------------------------------
    def _roots_tensor(self, roots) -> torch.Tensor:
        if isinstance(roots, torch.Tensor):
            r = roots.to(device=self.device)
            if r.dtype != torch.int32:
                r = r.to(torch.int64).to(torch.int32) if r.dtype != torch.int64 else r.to(torch.int32)
            return r.contiguous()
        a = np.asarray(roots, dtype=np.int64).astype(np.uint32).view(np.int32)
        return torch.from_numpy(a).to(self.device)

    def alloc_tree(self, b: int, fanouts: Sequence[int]) -> Tree:
        t = GiglTree()
        nbr, cnt, parents = [], [], b
        for k, f in enumerate(fanouts):
            cnt.append(torch.empty(max(parents, 1), dtype=torch.int32, device=self.device))
            parents *= int(f)
            nbr.append(torch.empty(max(parents, 1), dtype=torch.int32, device=self.device))
            t.nbr[k] = nbr[-1].data_ptr()
            t.cnt[k] = cnt[-1].data_ptr()
        return Tree(roots=None, fanouts=[int(f) for f in fanouts], nbr=nbr, cnt=cnt, c_struct=t)

    def sample_khop(self, roots, fanouts: Sequence[int], sampling_seed: int = 42,
                    mode: int = MODE_SPARK_HASH, out: Optional[Tree] = None) -> Tree:
        assert self._graph is not None, "load a graph first"
        r = self._roots_tensor(roots)
        b = int(r.numel())
        tree = out if out is not None else self.alloc_tree(b, fanouts)
        tree.roots = r
        fo = (C.c_int32 * len(fanouts))(*[int(f) for f in fanouts])
        check(self._lib.gigl_sample_khop(self._ctx, self._graph, C.c_void_p(r.data_ptr()), b, fo, len(fanouts),
                                         sampling_seed, mode, C.byref(tree.c_struct)), self._ctx)
        if b:  # trim the >=1 padding used for empty allocations
            parents = b
            for k, f in enumerate(tree.fanouts):
                tree.cnt[k] = tree.cnt[k][:parents]
                parents *= f
                tree.nbr[k] = tree.nbr[k][:parents]
        return tree

    def rows_dedup(self, ids: torch.Tensor) -> torch.Tensor:
        """in place: later occurrences of an id inside a row of `ids` [rows, width] (int32, uint32 payload) become
        GIGL_INVALID — the set union that forms a SamplingOp's input frontier (gigl_rows_dedup)"""
        assert ids.dim() == 2 and ids.is_contiguous() and ids.dtype == torch.int32 and ids.is_cuda
        check(self._lib.gigl_rows_dedup(self._ctx, C.c_void_p(ids.data_ptr()), ids.shape[0], ids.shape[1]), self._ctx)
        return ids

    def expand_frontier(self, nodes: torch.Tensor, ksums: torch.Tensor, f: int, hash_add: int, world: int,
                        max_window_end: int = -1, label_edges: Optional[str] = None):
        """one hop over an explicit frontier on this rank's shard (resident graph = rows of owned nodes), or on a
        named edge list loaded with load_label_edges (rows = its sources);
        nodes/ksums: int32 device tensors (uint32 payload).  -> (nbr [m*f] int32, cnt [m] int32)"""
        graph = self._label_edges[label_edges]["graph"] if label_edges else self._graph
        assert graph is not None
        m = int(nodes.numel())
        nbr = torch.empty(max(m * f, 1), dtype=torch.int32, device=self.device)
        cnt = torch.empty(max(m, 1), dtype=torch.int32, device=self.device)
        hash_add = ((int(hash_add) + 2**31) % 2**32) - 2**31
        check(self._lib.gigl_expand_frontier(self._ctx, graph, C.c_void_p(nodes.data_ptr()),
                                             C.c_void_p(ksums.data_ptr()), m, f, hash_add, world, max_window_end,
                                             C.c_void_p(nbr.data_ptr()), C.c_void_p(cnt.data_ptr())), self._ctx)
        return nbr[: m * f], cnt[:m]

    def frontier_bucket(self, nodes: torch.Tensor, ksums: Optional[torch.Tensor], world: int, cap: int,
                        req: torch.Tensor, slot_idx: torch.Tensor, counts: torch.Tensor) -> None:
        """bucket frontier slots by owner (gigl_frontier_bucket): req int32 [world, 2, cap], slot_idx int32
        [world, cap], counts int32 [world + 1] (last = overflow flag); all device tensors, no host sync"""
        m = int(nodes.numel())
        check(self._lib.gigl_frontier_bucket(self._ctx, C.c_void_p(nodes.data_ptr()),
                                             C.c_void_p(ksums.data_ptr()) if ksums is not None else None, m, world, cap,
                                             C.c_void_p(req.data_ptr()), C.c_void_p(slot_idx.data_ptr()),
                                             C.c_void_p(counts.data_ptr())), self._ctx)

    def frontier_scatter(self, resp: torch.Tensor, slot_idx: torch.Tensor, counts: torch.Tensor,
                         parent_ksums: torch.Tensor, m: int, world: int, cap: int, f: int, out_nbr: torch.Tensor,
                         out_cnt: torch.Tensor, child_ksums: Optional[torch.Tensor]) -> None:
        """owners' answers [world, cap, f] -> tree slots (gigl_frontier_scatter)"""
        check(self._lib.gigl_frontier_scatter(self._ctx, C.c_void_p(resp.data_ptr()), C.c_void_p(slot_idx.data_ptr()),
                                              C.c_void_p(counts.data_ptr()), C.c_void_p(parent_ksums.data_ptr()), m,
                                              world, cap, f, C.c_void_p(out_nbr.data_ptr()),
                                              C.c_void_p(out_cnt.data_ptr()),
                                              C.c_void_p(child_ksums.data_ptr()) if child_ksums is not None else None),
              self._ctx)

    def sample_positives(self, roots, num_positives: int, sampling_seed: int = 42, *, counter: int = 3,
                         label_edges: Optional[str] = None):
        """`num_positives` out-neighbours of every root (sampleDstNodesUniformly): of the main out-edge graph, or of a
        user-defined label edge list loaded with load_label_edges(label_edges, ...).  `counter` = the job's
        hashBasedUniformPermutation call number (3 for positives, 4 for user-defined hard negatives)."""
        g = self._label_edges[label_edges]["graph"] if label_edges else self._graph_out
        assert g is not None, "load the out-edge graph first (out_graph=True)"
        r = self._roots_tensor(roots)
        b = int(r.numel())
        pos = torch.empty(max(b * num_positives, 1), dtype=torch.int32, device=self.device)
        cnt = torch.empty(max(b, 1), dtype=torch.int32, device=self.device)
        check(self._lib.gigl_sample_out_neighbors(self._ctx, g, C.c_void_p(r.data_ptr()), b, num_positives,
                                                  sampling_seed, counter, MODE_SPARK_HASH, C.c_void_p(pos.data_ptr()),
                                                  C.c_void_p(cnt.data_ptr())), self._ctx)
        return pos[: b * num_positives], cnt[:b]

    def load_label_edges(self, name: str, n: int, src, dst, feats=None) -> None:
        """a user-defined label edge list ("pos" / "neg": loadEdgeDataframeIntoSparkSql with EdgeUsageType.POS / NEG —
        never bidirectionalised) as CSR by source, plus its feature rows put in that graph's `col` order (several
        input rows for one (src, dst): the first wins)"""
        if not hasattr(self, "_label_edges"):
            self._label_edges = {}
        old = self._label_edges.pop(name, None)
        if old:
            self._free_label_edges(old)
        s = np.ascontiguousarray(np.asarray(src).astype(np.uint32))
        d = np.ascontiguousarray(np.asarray(dst).astype(np.uint32))
        g = C.c_void_p()
        # roles swapped: rows = sources, columns = destinations
        check(self._lib.gigl_graph_build_from_coo(self._ctx, n, int(s.shape[0]), C.c_void_p(d.ctypes.data),
                                                  C.c_void_p(s.ctypes.data), LOC_HOST, 1, C.byref(g)), self._ctx)
        nn, ee = C.c_int64(), C.c_int64()
        check(self._lib.gigl_graph_info(g, C.byref(nn), C.byref(ee)), self._ctx)
        entry = {"graph": g, "feat": None, "table": None, "n_edges": ee.value}
        if feats is not None and np.asarray(feats).size:
            f = torch.from_numpy(np.ascontiguousarray(np.asarray(feats, dtype=np.float32))).to(self.device)
            assert f.dim() == 2 and f.shape[0] == s.shape[0]
            st = torch.from_numpy(s.view(np.int32)).to(self.device)
            dt = torch.from_numpy(d.view(np.int32)).to(self.device)
            eid = torch.empty(st.numel(), dtype=torch.int64, device=self.device)
            check(self._lib.gigl_edge_ids(self._ctx, g, C.c_void_p(dt.data_ptr()), C.c_void_p(st.data_ptr()), st.numel(),
                                          C.c_void_p(eid.data_ptr())), self._ctx)
            self._stream.synchronize()
            win = torch.full((ee.value,), st.numel(), dtype=torch.int64, device=self.device)
            win.scatter_reduce_(0, eid, torch.arange(st.numel(), device=self.device), reduce="amin")
            table = f[win].contiguous()
            h = C.c_void_p()
            check(self._lib.gigl_features_load(self._ctx, table.shape[0], table.shape[1], DTYPE_F32,
                                               C.c_void_p(table.data_ptr()), LOC_DEVICE, C.byref(h)), self._ctx)
            entry["feat"], entry["table"] = h, table
        self._label_edges[name] = entry

    def _free_label_edges(self, entry) -> None:
        if entry.get("feat"):
            self._lib.gigl_features_destroy(entry["feat"])
        if entry.get("graph"):
            self._lib.gigl_graph_destroy(entry["graph"])

    def encode_typed_records(self, roots: torch.Tensor, root_node_type: int, ops, feats, *, tfrecord_frame: bool = True,
                             edge_feats=None, kind: int = _lib.REC_ROOTED_NODE_NEIGHBORHOOD):
        """typed (heterogeneous) RootedNodeNeighborhood records — or, kind REC_NODE_ANCHOR_LINK_PRED, the typed
        NodeAnchorBasedLinkPredictionSample records — on the device (gigl_typed_samples_encode).
        roots: int32 [b] on the device; ops: sequence of (frontier [b, w], nbr [b, w, f], condensed_edge_type,
        result_node_type, outgoing[, positive]) — positive: the op sampled the roots' positive edges (pos_edges);
        feats: per condensed node type a float32 [n, d] device tensor or None;
        edge_feats: per condensed edge type the name of a load_label_edges entry (the type's edges as CSR by source with
        their feature rows) or None.
        -> (uint8 device tensor of all records back to back, int64 device tensor rec_off[b + 1])"""
        b = int(roots.numel())
        roots = roots.to(device=self.device, dtype=torch.int32).contiguous()
        c_ops = (_lib.GiglTypedOp * len(ops))()
        keep = [roots]
        for i, op in enumerate(ops):
            front, nbr, cet, res_t, outgoing = op[:5]
            positive = bool(op[5]) if len(op) > 5 else False
            front = front.to(device=self.device, dtype=torch.int32).contiguous()
            nbr = nbr.to(device=self.device, dtype=torch.int32).contiguous()
            keep += [front, nbr]
            w = int(front.numel() // max(b, 1))
            f = int(nbr.numel() // max(b * w, 1))
            c_ops[i].frontier, c_ops[i].nbr = front.data_ptr(), nbr.data_ptr()
            c_ops[i].w, c_ops[i].f = w, f
            c_ops[i].condensed_edge_type, c_ops[i].result_node_type = int(cet), int(res_t)
            c_ops[i].outgoing = (1 if outgoing else 0) | (2 if positive else 0)  # GIGL_TYPED_OP_POSITIVE
        c_feats = (_lib.GiglTypedFeat * len(feats))()
        for t, x in enumerate(feats):
            if x is None:
                continue
            x = x.to(device=self.device, dtype=torch.float32).contiguous()
            keep.append(x)
            c_feats[t].x, c_feats[t].d, c_feats[t].n = x.data_ptr(), int(x.shape[1]), int(x.shape[0])
        edge_feats = list(edge_feats or [])
        c_ef = (_lib.GiglTypedEdgeFeat * max(len(edge_feats), 1))()
        for t, name in enumerate(edge_feats):
            if name is None:
                continue
            e = self._label_edges[name]
            if e["table"] is None:
                continue
            c_ef[t].by_source, c_ef[t].feat, c_ef[t].d = e["graph"], e["table"].data_ptr(), int(e["table"].shape[1])
        cap = C.c_int64()
        rc = self._lib.gigl_typed_records_capacity(c_ops, len(ops), c_feats, len(feats), c_ef, len(edge_feats), b,
                                                   1 if tfrecord_frame else 0, C.byref(cap))
        if rc != 0:
            raise ValueError("gigl_typed_records_capacity: between 1 and 16 ops / node types")
        out = torch.empty(max(cap.value, 1), dtype=torch.uint8, device=self.device)
        rec_off = torch.empty(b + 1, dtype=torch.int64, device=self.device)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()  # the op results may come from torch's stream
        check(self._lib.gigl_typed_samples_encode(self._ctx, int(kind), C.c_void_p(roots.data_ptr()), int(root_node_type),
                                                  c_ops, len(ops), c_feats, len(feats), c_ef, len(edge_feats), b,
                                                  1 if tfrecord_frame else 0,
                                                  C.c_void_p(out.data_ptr()), cap.value,
                                                  C.c_void_p(rec_off.data_ptr()), C.c_void_p(status.data_ptr())),
              self._ctx)
        self._stream.synchronize()
        if int(status.item()) != 0:
            raise RuntimeError("gigl_typed_samples_encode: output capacity too small (status=1)")
        del keep
        return out[: int(rec_off[-1].item())], rec_off

    def encode_records(self, tree: Tree, *, kind: int = _lib.REC_ROOTED_NODE_NEIGHBORHOOD, trees_per_record: int = 1,
                       condensed_node_type: Optional[int] = 0, condensed_edge_type: Optional[int] = 0,
                       tfrecord_frame: bool = True, emit: Optional[torch.Tensor] = None,
                       suffix: Optional[torch.Tensor] = None, suffix_off: Optional[torch.Tensor] = None,
                       with_features: bool = True, with_edge_features: bool = True, n_neg_trees: int = 0,
                       pos_label_edges: Optional[str] = None, neg_label_edges: Optional[str] = None):
        """sampled trees -> serialized RootedNodeNeighborhood / SupervisedNodeClassificationSample (suffix = encoded
        labels) / NodeAnchorBasedLinkPredictionSample records, encoded on the device (gigl_records_encode).
        -> (uint8 device tensor of all records back to back, int64 device tensor rec_off[n_records+1])"""
        assert tree.roots is not None and tree.b % trees_per_record == 0
        n_rec = tree.b // trees_per_record
        o = _lib.GiglRecordOpts()
        o.kind, o.trees_per_record = kind, trees_per_record
        o.condensed_node_type = -1 if condensed_node_type is None else int(condensed_node_type)
        o.condensed_edge_type = -1 if condensed_edge_type is None else int(condensed_edge_type)
        o.tfrecord_frame = 1 if tfrecord_frame else 0
        keep = []
        if emit is not None:
            emit = emit.to(device=self.device, dtype=torch.uint8).contiguous()
            assert emit.numel() == n_rec
            o.emit = emit.data_ptr()
            keep.append(emit)
        suffix_total = 0
        if suffix is not None:
            suffix = suffix.to(device=self.device, dtype=torch.uint8).contiguous()
            suffix_off = suffix_off.to(device=self.device, dtype=torch.int64).contiguous()
            assert suffix_off.numel() == n_rec + 1
            suffix_total = int(suffix.numel())
            if suffix_total == 0:  # data_ptr() of an empty tensor is NULL
                suffix = torch.zeros(1, dtype=torch.uint8, device=self.device)
            o.suffix, o.suffix_off = suffix.data_ptr(), suffix_off.data_ptr()
            keep += [suffix, suffix_off]
        feat = self._feat if with_features else None
        if with_edge_features and getattr(self, "_efeat_handle", None):
            o.graph, o.edge_feat = self._graph, self._efeat_handle  # Edge.feature_values from the resident table
        o.n_neg_trees = int(n_neg_trees)
        if pos_label_edges:  # user-defined label edges: their own edge list (and feature table)
            e = self._label_edges[pos_label_edges]
            o.pos_edges_graph, o.pos_edge_feat = e["graph"], e["feat"]
        if neg_label_edges:
            e = self._label_edges[neg_label_edges]
            o.neg_edges_graph, o.neg_edge_feat = e["graph"], e["feat"]
        fo = (C.c_int32 * len(tree.fanouts))(*tree.fanouts)
        cap = C.c_int64()
        check(self._lib.gigl_records_capacity(fo, len(tree.fanouts), self.feat_dim if feat else 0, C.byref(o), n_rec,
                                              suffix_total, C.byref(cap)))
        out = torch.empty(max(cap.value, 1), dtype=torch.uint8, device=self.device)
        rec_off = torch.empty(n_rec + 1, dtype=torch.int64, device=self.device)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        tree.c_struct.hops, tree.c_struct.b = len(tree.fanouts), tree.b
        for k, f in enumerate(tree.fanouts):
            tree.c_struct.fanouts[k] = f
        check(self._lib.gigl_records_encode(self._ctx, C.c_void_p(tree.roots.data_ptr()), C.byref(tree.c_struct), feat,
                                            C.byref(o), n_rec, C.c_void_p(out.data_ptr()), cap.value,
                                            C.c_void_p(rec_off.data_ptr()), C.c_void_p(status.data_ptr())), self._ctx)
        self._stream.synchronize()
        if int(status.item()) != 0:
            raise RuntimeError("gigl_records_encode: output capacity too small (status=1)")
        del keep
        return out[: int(rec_off[-1].item())], rec_off

    def encode_avro_embeddings_async(self, ids: torch.Tensor, emb: torch.Tensor, node_type: str, sync_marker: bytes):
        """enqueue the encoding of (ids [n], emb [n, D]) as Avro data blocks of `Embedding` records on this engine's
        stream (gigl_avro_embeddings_encode) without waiting for it.
        -> (out uint8 [capacity] device, rec_off int64 [n] device, scal int64 [2] device: total bytes | status)"""
        assert len(sync_marker) == 16 and ids.dim() == 1 and emb.dim() == 2 and emb.shape[0] == ids.numel()
        if not ids.is_cuda:  # (a pageable upload would wait for the stream: stage through pinned memory)
            ids = ids.to(torch.int64).pin_memory().to(self.device, non_blocking=True)
        ids = ids.to(device=self.device, dtype=torch.int64).contiguous()
        emb = emb.to(device=self.device, dtype=torch.float32)
        if emb.stride(1) != 1 and emb.numel():
            emb = emb.contiguous()
        n, d = int(ids.numel()), int(emb.shape[1])
        stride = int(emb.stride(0)) if n > 1 else d
        ty = node_type.encode("utf-8")
        per, nblk, cap = C.c_int32(), C.c_int64(), C.c_int64()
        rc = self._lib.gigl_avro_embeddings_layout(n, d, len(ty), C.byref(per), C.byref(nblk), C.byref(cap))
        if rc != 0:
            raise ValueError(f"node type of {len(ty)} UTF-8 bytes is too long for the device encoder")
        out = torch.empty(max(cap.value, 1), dtype=torch.uint8, device=self.device)
        rec_off = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        scal = torch.zeros(2, dtype=torch.int64, device=self.device)  # total bytes | status (int32 in the low half)
        check(self._lib.gigl_avro_embeddings_encode(
            self._ctx, C.c_void_p(ids.data_ptr()), C.c_void_p(emb.data_ptr()), max(stride, d), n, d, ty, len(ty),
            bytes(sync_marker), C.c_void_p(out.data_ptr()), cap.value, C.c_void_p(rec_off.data_ptr()),
            C.c_void_p(scal.data_ptr()), C.c_void_p(scal.data_ptr() + 8)), self._ctx)
        for t in (ids, emb):  # inputs stay alive until the stream has consumed them
            t.record_stream(self._stream)
        return out, rec_off[:n], scal

    def encode_avro_embeddings(self, ids: torch.Tensor, emb: torch.Tensor, node_type: str, sync_marker: bytes):
        """(ids [n], emb [n, D]) -> Avro data blocks of `Embedding` records, encoded on the device
        (gigl_avro_embeddings_encode).  -> (uint8 device tensor of the blocks, int64 device tensor rec_off[n])"""
        out, rec_off, scal = self.encode_avro_embeddings_async(ids, emb, node_type, sync_marker)
        self._stream.synchronize()
        total, status = (int(v) for v in scal.tolist())
        if status & 0xFFFFFFFF:
            raise RuntimeError("gigl_avro_embeddings_encode: output capacity too small (status=1)")
        return out[:total], rec_off

    def union_capacity(self, b: int, fanouts: Sequence[int]):
        fo = (C.c_int32 * len(fanouts))(*[int(f) for f in fanouts])
        cn, ce = C.c_int64(), C.c_int64()
        check(self._lib.gigl_union_capacity(b, fo, len(fanouts), C.byref(cn), C.byref(ce)))
        return cn.value, ce.value

    def alloc_union(self, b: int, fanouts: Sequence[int]) -> UnionGraph:
        cn, ce = self.union_capacity(b, fanouts)
        dev = self.device
        u = GiglUnion()
        # (one zero fill for the three arrays that start cleared; each slice starts on a 16-byte boundary)
        pad = lambda k: (k + 3) // 4 * 4
        z = torch.zeros(pad(GIGL_META_LEN) + 2 * pad(cn + 2), dtype=torch.int32, device=dev)
        meta = z[:GIGL_META_LEN]
        rowptr = z[pad(GIGL_META_LEN): pad(GIGL_META_LEN) + cn + 2]
        rowend = z[pad(GIGL_META_LEN) + pad(cn + 2): pad(GIGL_META_LEN) + pad(cn + 2) + cn + 2]
        nodes = torch.empty(max(cn, 1), dtype=torch.int32, device=dev)
        col = torch.empty(max(ce, 1), dtype=torch.int32, device=dev)
        root_local = torch.empty(max(b, 1), dtype=torch.int32, device=dev)
        u.meta, u.nodes, u.rowptr, u.rowend, u.col, u.root_local = (
            meta.data_ptr(), nodes.data_ptr(), rowptr.data_ptr(), rowend.data_ptr(), col.data_ptr(),
            root_local.data_ptr())
        u.cap_nodes, u.cap_edges = cn, ce
        return UnionGraph(meta=meta, nodes=nodes, rowptr=rowptr, rowend=rowend, col=col, root_local=root_local,
                          hops=len(fanouts), c_struct=u)

    def union_build(self, tree: Tree, out: Optional[UnionGraph] = None, group_roots: Optional[int] = None
                    ) -> UnionGraph:
        """group_roots: treat the tree's roots as consecutive independent batches of that many roots
        (gigl_union_build_groups): dedup within a batch only, one level-ordered numbering over all of them"""
        u = out if out is not None else self.alloc_union(tree.b, tree.fanouts)
        if group_roots is None or group_roots == tree.b:
            check(self._lib.gigl_union_build(self._ctx, C.c_void_p(tree.roots.data_ptr()), C.byref(tree.c_struct),
                                             C.byref(u.c_struct)), self._ctx)
        else:
            check(self._lib.gigl_union_build_groups(self._ctx, C.c_void_p(tree.roots.data_ptr()),
                                                    C.byref(tree.c_struct), int(group_roots), C.byref(u.c_struct)),
                  self._ctx)
        return u

    def gather_mean(self, src: Optional[torch.Tensor], d: int, gather_ids: Optional[torch.Tensor],
                    rowptr: torch.Tensor, rowend: Optional[torch.Tensor], col: torch.Tensor,
                    n_rows_dev: torch.Tensor, rows_cap: int, out: Optional[torch.Tensor] = None,
                    aggr: str = "mean") -> torch.Tensor:
        """src None -> the resident feature table; rowend None -> packed CSR (rowend = rowptr[1:]).
        aggr: "mean" | "sum" | "max" (PyG SAGEConv aggr)."""
        if rowend is None:
            rowend = rowptr[1:]
        if src is None:
            src_ptr, dt = self._feat_ptr, self.feat_dtype
        else:
            assert src.is_cuda and src.is_contiguous()
            src_ptr = C.c_void_p(src.data_ptr())
            dt = DTYPE_F32 if src.dtype == torch.float32 else DTYPE_F16
        if out is None:
            out = torch.empty((rows_cap, 2 * d), dtype=torch.float32, device=self.device)
        gid = C.c_void_p(gather_ids.data_ptr()) if gather_ids is not None else None
        check(self._lib.gigl_gather_reduce(self._ctx, src_ptr, dt, d, gid, C.c_void_p(rowptr.data_ptr()),
                                           C.c_void_p(rowend.data_ptr()), C.c_void_p(col.data_ptr()),
                                           C.c_void_p(n_rows_dev.data_ptr()), rows_cap, _lib.AGGR[aggr],
                                           C.c_void_p(out.data_ptr())), self._ctx)
        return out

    def gather_rows(self, ids: torch.Tensor, n_dev: torch.Tensor, cap: int,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """fp32 rows of the resident feature table for the first *n_dev ids"""
        if out is None:
            out = torch.empty((cap, self.feat_dim), dtype=torch.float32, device=self.device)
        check(self._lib.gigl_gather_rows(self._ctx, self._feat_ptr, self.feat_dtype, self.feat_dim,
                                         C.c_void_p(ids.data_ptr()), C.c_void_p(n_dev.data_ptr()), cap,
                                         C.c_void_p(out.data_ptr())), self._ctx)
        return out

    def gcn_aggregate(self, h: Optional[torch.Tensor], d: int, gather_ids: Optional[torch.Tensor], u: "UnionGraph",
                      n_rows_dev: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """GCN-normalised aggregation over the union graph `u` (h None -> resident feature table)"""
        cap = int(u.nodes.numel())
        if h is None:
            h_ptr, dt = self._feat_ptr, self.feat_dtype
        else:
            assert h.is_cuda and h.is_contiguous()
            h_ptr, dt = C.c_void_p(h.data_ptr()), (DTYPE_F32 if h.dtype == torch.float32 else DTYPE_F16)
        if out is None:
            out = torch.empty((cap, d), dtype=torch.float32, device=self.device)
        dinv = torch.empty(cap, dtype=torch.float32, device=self.device)
        gid = C.c_void_p(gather_ids.data_ptr()) if gather_ids is not None else None
        check(self._lib.gigl_gcn_aggregate(self._ctx, h_ptr, dt, d, gid, C.c_void_p(u.rowptr.data_ptr()),
                                           C.c_void_p(u.rowend.data_ptr()), C.c_void_p(u.col.data_ptr()),
                                           C.c_void_p(u.meta.data_ptr()), cap, C.c_void_p(n_rows_dev.data_ptr()), cap,
                                           C.c_void_p(bias.data_ptr()) if bias is not None else None, act,
                                           C.c_void_p(dinv.data_ptr()), C.c_void_p(out.data_ptr())), self._ctx)
        return out

    def gat_aggregate(self, h: torch.Tensor, att_src: torch.Tensor, att_dst: torch.Tensor, heads: int, channels: int,
                      u: "UnionGraph", n_rows_dev: torch.Tensor, bias: Optional[torch.Tensor], concat: bool = True,
                      negative_slope: float = 0.2, act: int = 0, out: Optional[torch.Tensor] = None,
                      edge_attr: Optional[torch.Tensor] = None, att_edge_folded: Optional[torch.Tensor] = None,
                      w_edge_msg: Optional[torch.Tensor] = None) -> torch.Tensor:
        """edge_attr: [cap_edges, De] rows aligned with u.col (union_edge_attr); att_edge_folded: [heads, De];
        w_edge_msg: [heads*channels, De] adds W e_ij to every message (EdgeAttrGATConv)"""
        cap = int(u.nodes.numel())
        assert h.is_cuda and h.is_contiguous() and h.dtype == torch.float32 and h.shape[1] == heads * channels
        if out is None:
            out = torch.empty((cap, heads * channels if concat else channels), dtype=torch.float32, device=self.device)
        common = (self._ctx, C.c_void_p(h.data_ptr()), C.c_void_p(att_src.data_ptr()), C.c_void_p(att_dst.data_ptr()),
                  heads, channels, negative_slope, 1 if concat else 0, C.c_void_p(u.rowptr.data_ptr()),
                  C.c_void_p(u.rowend.data_ptr()), C.c_void_p(u.col.data_ptr()), C.c_void_p(u.meta.data_ptr()), cap,
                  C.c_void_p(n_rows_dev.data_ptr()), cap, C.c_void_p(bias.data_ptr()) if bias is not None else None, act)
        if edge_attr is None:
            scratch = torch.empty(2 * cap * heads, dtype=torch.float32, device=self.device)
            check(self._lib.gigl_gat_aggregate(*common, C.c_void_p(scratch.data_ptr()), C.c_void_p(out.data_ptr())),
                  self._ctx)
            return out
        ce = int(u.col.numel())
        de = int(edge_attr.shape[1])
        assert edge_attr.is_contiguous() and edge_attr.dtype == torch.float32 and edge_attr.shape[0] == ce
        assert att_edge_folded.is_contiguous() and tuple(att_edge_folded.shape) == (heads, de)
        if w_edge_msg is not None:
            assert w_edge_msg.is_contiguous() and tuple(w_edge_msg.shape) == (heads * channels, de)
        scratch = torch.empty(2 * cap * heads + ce * heads, dtype=torch.float32, device=self.device)
        check(self._lib.gigl_gat_aggregate_edge(
            *common, C.c_void_p(edge_attr.data_ptr()), de, ce, C.c_void_p(att_edge_folded.data_ptr()),
            C.c_void_p(w_edge_msg.data_ptr()) if w_edge_msg is not None else None, C.c_void_p(scratch.data_ptr()),
            C.c_void_p(out.data_ptr())), self._ctx)
        return out

    def gat_input_layer(self, ids: torch.Tensor, w: torch.Tensor, att_src: torch.Tensor, att_dst: torch.Tensor,
                        heads: int, channels: int, u: "UnionGraph", n_src_dev: torch.Tensor, n_rows_dev: torch.Tensor,
                        bias: Optional[torch.Tensor], negative_slope: float = 0.2, act: int = 0) -> Optional[torch.Tensor]:
        """first GATConv layer over the resident feature table from the input side (gigl_gat_input_layer): logits
        from folded attention vectors, projection after the aggregation.  None when the shape is outside the built
        ones (the caller takes gather_rows + linear + gat_aggregate)."""
        d, cap = self.feat_dim, int(u.nodes.numel())
        if d % 4 or heads not in (1, 2, 4) or self.feat_dtype not in (DTYPE_F32, DTYPE_F16):
            return None
        assert w.is_contiguous() and tuple(w.shape) == (heads * channels, d) and w.dtype == torch.float32
        ce = int(u.col.numel())
        n_scr = int(self._lib.gigl_gat_input_layer_scratch(d, heads, cap, cap, ce))
        scratch = torch.empty(n_scr, dtype=torch.float32, device=self.device)
        out = torch.empty((cap, heads * channels), dtype=torch.float32, device=self.device)
        check(self._lib.gigl_gat_input_layer(
            self._ctx, self._feat_ptr, self.feat_dtype, d, C.c_void_p(ids.data_ptr()), C.c_void_p(w.data_ptr()),
            C.c_void_p(att_src.data_ptr()), C.c_void_p(att_dst.data_ptr()), heads, channels, negative_slope,
            C.c_void_p(u.rowptr.data_ptr()), C.c_void_p(u.rowend.data_ptr()), C.c_void_p(u.col.data_ptr()),
            ce, C.c_void_p(n_src_dev.data_ptr()), cap, C.c_void_p(n_rows_dev.data_ptr()), cap,
            C.c_void_p(bias.data_ptr()) if bias is not None else None, act, C.c_void_p(scratch.data_ptr()),
            C.c_void_p(out.data_ptr())), self._ctx)
        return out

    def gine_aggregate(self, x: torch.Tensor, edge_rows: torch.Tensor, eps: torch.Tensor, u, n_rows_dev: torch.Tensor
                       ) -> torch.Tensor:
        """(1 + eps) x_i + sum_e relu(x_j + edge_rows_e) over the CSR view `u` (gigl_gine_aggregate): [rows, d]"""
        rows, d = int(x.shape[0]), int(x.shape[1])
        assert x.is_cuda and x.is_contiguous() and edge_rows.is_contiguous() and edge_rows.shape[1] == d
        out = torch.zeros((rows, d), dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(self._lib.gigl_gine_aggregate(self._ctx, p(x), p(edge_rows), p(eps), d, p(u.rowptr), p(u.rowend), p(u.col),
                                            p(n_rows_dev), rows, p(out)), self._ctx)
        return out

    def gine_aggregate_backward(self, x, edge_rows, eps, u, n_rows_dev, dout):
        """-> (dx, dedge_rows, deps): gigl_gine_aggregate_backward"""
        dout = dout.contiguous()
        dx, dee, deps = torch.zeros_like(x), torch.zeros_like(edge_rows), torch.zeros(1, dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(self._lib.gigl_gine_aggregate_backward(self._ctx, p(x), p(edge_rows), p(eps), int(x.shape[1]), p(u.rowptr),
                                                     p(u.rowend), p(u.col), p(n_rows_dev), int(x.shape[0]), p(dout), p(dx),
                                                     p(dee), p(deps)), self._ctx)
        return dx, dee, deps

    def transformer_aggregate_edge(self, q, k, v, edge_rows, heads: int, channels: int, u, n_rows_dev) -> torch.Tensor:
        """TransformerConv attention with edge rows (gigl_transformer_aggregate_edge) over the CSR view `u`"""
        rows = int(q.shape[0])
        out = torch.zeros((rows, heads * channels), dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(self._lib.gigl_transformer_aggregate_edge(self._ctx, p(q), p(k), p(v), p(edge_rows), heads, channels,
                                                        p(u.rowptr), p(u.rowend), p(u.col), p(n_rows_dev), rows, p(out)),
              self._ctx)
        return out

    def transformer_aggregate_edge_backward(self, q, k, v, edge_rows, heads: int, channels: int, u, n_rows_dev, out, dout):
        """-> (dq, dk, dv, dedge_rows)"""
        dout = dout.contiguous()
        dq, dk, dv, dxe = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v), torch.zeros_like(edge_rows)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(self._lib.gigl_transformer_aggregate_edge_backward(
            self._ctx, p(q), p(k), p(v), p(edge_rows), heads, channels, p(u.rowptr), p(u.rowend), p(u.col), p(n_rows_dev),
            int(q.shape[0]), p(out), p(dout), p(dq), p(dk), p(dv), p(dxe)), self._ctx)
        return dq, dk, dv, dxe

    def gatv2_aggregate(self, xl: torch.Tensor, xr: torch.Tensor, att: torch.Tensor, heads: int, channels: int, u,
                        n_rows_dev: torch.Tensor, bias: Optional[torch.Tensor], negative_slope: float = 0.2,
                        act: int = 0, edge_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """GATv2Conv attention + aggregation over the CSR view `u` (rowptr / rowend / col): [rows, heads*channels];
        edge_rows: lin_edge(edge_attr) [edges, heads*channels] in `col` order (GATv2Conv(edge_dim))"""
        rows = int(xr.shape[0])
        for t in (xl, xr):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.shape[1] == heads * channels
        out = torch.empty((rows, heads * channels), dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        if edge_rows is not None:
            assert edge_rows.is_contiguous() and tuple(edge_rows.shape) == (int(u.col.numel()), heads * channels)
        check(self._lib.gigl_gatv2_aggregate_edge(self._ctx, p(xl), p(xr), p(att), heads, channels, negative_slope,
                                                  p(u.rowptr), p(u.rowend), p(u.col), p(n_rows_dev), rows, p(bias), act,
                                                  p(edge_rows), p(out)), self._ctx)
        return out

    def gatv2_aggregate_backward(self, xl, xr, att, heads: int, channels: int, u, n_rows_dev, out_pre, dout,
                                 negative_slope: float = 0.2, edge_rows: Optional[torch.Tensor] = None):
        """-> (dxl, dxr, datt, dedge_rows | None): gigl_gatv2_aggregate_edge_backward"""
        dout = dout.contiguous()
        dxl, dxr, datt = torch.zeros_like(xl), torch.zeros_like(xr), torch.zeros_like(att)
        dxe = torch.zeros_like(edge_rows) if edge_rows is not None else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(self._lib.gigl_gatv2_aggregate_edge_backward(
            self._ctx, p(xl), p(xr), p(att), heads, channels, negative_slope, p(u.rowptr), p(u.rowend), p(u.col),
            p(n_rows_dev), int(xr.shape[0]), p(out_pre), p(dout), p(edge_rows), p(dxl), p(dxr), p(datt), p(dxe)), self._ctx)
        return dxl, dxr, datt, dxe

    def gat_input_layer_fused(self, ids: torch.Tensor, w: torch.Tensor, att_src: torch.Tensor, att_dst: torch.Tensor,
                              heads: int, channels: int, u: "UnionGraph", n_rows_dev: torch.Tensor,
                              bias: Optional[torch.Tensor], negative_slope: float = 0.2, act: int = 0,
                              n_local_dev: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """gigl_gat_input_layer_fused (one pass, logits formed on the fly); None outside the built shapes"""
        d, cap = self.feat_dim, int(u.nodes.numel())
        if d % 4 or d > 1024 or heads not in (1, 2, 4) or self.feat_dtype not in (DTYPE_F32, DTYPE_F16):
            return None
        n_scr = int(self._lib.gigl_gat_input_layer_fused_scratch(d, heads, cap))
        scratch = torch.empty(n_scr, dtype=torch.float32, device=self.device)
        out = torch.empty((cap, heads * channels), dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(self._lib.gigl_gat_input_layer_fused(
            self._ctx, self._feat_ptr, self.feat_dtype, d, p(ids), p(n_local_dev), p(w), p(att_src), p(att_dst), heads,
            channels, negative_slope, p(u.rowptr), p(u.rowend), p(u.col), p(n_rows_dev), cap, p(bias), act, p(scratch),
            p(out)), self._ctx)
        return out

    def gat_aggregate_backward(self, h: torch.Tensor, att_src: torch.Tensor, att_dst: torch.Tensor, heads: int,
                               channels: int, u, n_rows_dev: torch.Tensor, out_pre: torch.Tensor, dout: torch.Tensor,
                               negative_slope: float = 0.2, edge_attr: Optional[torch.Tensor] = None,
                               att_edge_folded: Optional[torch.Tensor] = None, u_msg: Optional[torch.Tensor] = None):
        """-> (dh [cap, H*C], d_alpha_src [cap, H], d_alpha_dst [cap, H], d_alpha_edge [cap_edges, H] | None,
        z [cap, H, De] | None): the edge-wise part of the GAT layer's backward (gigl_gat_aggregate_backward); u_msg
        [cap, H, De] = W_msg^T dout per head switches the EdgeAttrGATConv message term on; see include/gigl_hip.h"""
        cap, ce = int(u.nodes.numel()), int(u.col.numel())
        hc = heads * channels
        for t in (h, out_pre, dout):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.shape[1] == hc
        dev = self.device
        dh = torch.zeros((cap, hc), dtype=torch.float32, device=dev)
        ds = torch.zeros((cap, heads), dtype=torch.float32, device=dev)
        dd = torch.zeros((cap, heads), dtype=torch.float32, device=dev)
        dae = torch.zeros((ce, heads), dtype=torch.float32, device=dev) if edge_attr is not None else None
        z = None
        if u_msg is not None:
            assert u_msg.is_contiguous() and tuple(u_msg.shape) == (cap, heads, int(edge_attr.shape[1]))
            z = torch.zeros_like(u_msg)
        scratch = torch.empty(2 * cap * heads + (ce * heads if edge_attr is not None else 0), dtype=torch.float32,
                              device=dev)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(self._lib.gigl_gat_aggregate_backward(
            self._ctx, p(h), p(att_src), p(att_dst), heads, channels, negative_slope, p(u.rowptr), p(u.rowend),
            p(u.col), p(u.meta), cap, p(n_rows_dev), cap, p(out_pre), p(dout), p(edge_attr),
            int(edge_attr.shape[1]) if edge_attr is not None else 0, ce, p(att_edge_folded), p(scratch), p(dh), p(ds),
            p(dd), p(dae), p(u_msg), p(z)), self._ctx)
        return dh, ds, dd, dae, z

    def gat_input_aggregate(self, u_fold: torch.Tensor, heads: int, rowptr: torch.Tensor, rowend: torch.Tensor,
                            col: torch.Tensor, node_ids: torch.Tensor, n_rows_dev: torch.Tensor, rows_cap: int,
                            negative_slope: float = 0.2) -> torch.Tensor:
        """z [heads, rows_cap, d]: the attention-weighted sums of the stored rows for the first rows_cap rows of a staged
        batch graph (gigl_gat_input_aggregate; u_fold [2*heads, d] = the folded attention vectors)"""
        d = self.feat_dim
        assert u_fold.is_cuda and u_fold.is_contiguous() and tuple(u_fold.shape) == (2 * heads, d)
        z = torch.empty((heads, rows_cap, d), dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(self._lib.gigl_gat_input_aggregate(self._ctx, self._feat_ptr, self.feat_dtype, d, p(node_ids), p(u_fold), heads,
                                                 negative_slope, p(rowptr), p(rowend), p(col), p(n_rows_dev), rows_cap,
                                                 p(z)), self._ctx)
        return z

    def gat_input_aggregate_backward(self, u_fold: torch.Tensor, heads: int, rowptr: torch.Tensor, rowend: torch.Tensor,
                                     col: torch.Tensor, node_ids: torch.Tensor, n_rows_dev: torch.Tensor, rows_cap: int,
                                     dz: torch.Tensor, negative_slope: float = 0.2) -> torch.Tensor:
        """du [2*heads, d] = the gradient of gat_input_aggregate w.r.t. the folded vectors, given dz [heads, rows_cap, d]"""
        d = self.feat_dim
        assert dz.is_cuda and dz.is_contiguous() and tuple(dz.shape) == (heads, rows_cap, d) and dz.dtype == torch.float32
        du = torch.zeros((2 * heads, d), dtype=torch.float32, device=self.device)
        scr = torch.empty(2 * heads * max(int(col.numel()), 1), dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(self._lib.gigl_gat_input_aggregate_backward(self._ctx, self._feat_ptr, self.feat_dtype, d, p(node_ids),
                                                          p(u_fold), heads, negative_slope, p(rowptr), p(rowend), p(col),
                                                          p(n_rows_dev), rows_cap, p(dz), p(scr), p(du)), self._ctx)
        return du

    def gat_backward_epilogue(self, dh: torch.Tensor, ds: torch.Tensor, dd: torch.Tensor, xw: torch.Tensor,
                              att_src: torch.Tensor, att_dst: torch.Tensor, heads: int, channels: int,
                              n_nodes_dev: torch.Tensor):
        """-> (dxw = dh, updated in place; d_att_src [H*C]; d_att_dst [H*C]): the dense tail of the GAT layer's backward
        in one pass (gigl_gat_backward_epilogue)"""
        hc = heads * channels
        for t in (dh, xw):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.shape[1] == hc
        assert ds.is_contiguous() and dd.is_contiguous() and ds.shape == dd.shape == (dh.shape[0], heads)
        a_s = att_src.detach().reshape(-1).to(torch.float32).contiguous()
        a_d = att_dst.detach().reshape(-1).to(torch.float32).contiguous()
        g_s = torch.zeros(hc, dtype=torch.float32, device=self.device)
        g_d = torch.zeros(hc, dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(self._lib.gigl_gat_backward_epilogue(self._ctx, p(dh), p(ds), p(dd), p(xw), p(a_s), p(a_d), p(n_nodes_dev),
                                                   int(dh.shape[0]), heads, channels, p(g_s), p(g_d)), self._ctx)
        return dh, g_s, g_d

    def gather_mean_backward(self, dout: torch.Tensor, d: int, rowptr: torch.Tensor, rowend: Optional[torch.Tensor],
                             col: torch.Tensor, n_rows_dev: torch.Tensor, rows_cap: int, dsrc: torch.Tensor,
                             aggr: str = "mean", src: Optional[torch.Tensor] = None) -> None:
        """dsrc (zero-filled, [n_src, d] fp32) += the gradient of gather_mean w.r.t. its dense local source
        (aggr "max" needs the forward's source matrix `src`)"""
        if rowend is None:
            rowend = rowptr[1:]
        assert dout.is_cuda and dout.is_contiguous() and dsrc.is_contiguous() and dout.dtype == torch.float32
        sp = C.c_void_p(src.data_ptr()) if src is not None else None
        check(self._lib.gigl_gather_reduce_backward(self._ctx, C.c_void_p(dout.data_ptr()), d,
                                                    C.c_void_p(rowptr.data_ptr()), C.c_void_p(rowend.data_ptr()),
                                                    C.c_void_p(col.data_ptr()), C.c_void_p(n_rows_dev.data_ptr()),
                                                    rows_cap, _lib.AGGR[aggr], sp, C.c_void_p(dsrc.data_ptr())),
              self._ctx)

    def linear_grouped(self, groups, k: int, n: int, act: int = 0) -> None:
        """groups: [(a [m_cap, k], w [n, k], bias [n] | None, m_dev int32 [1], y [m_cap, n])] of device fp32 tensors — every
        product y = act(a w^T + bias) over its first *m_dev rows, all in ONE launch (gigl_linear_grouped)"""
        import struct
        blob, cap = b"", 0
        for a, w, bias, m_dev, y in groups:
            assert a.is_contiguous() and w.is_contiguous() and y.is_contiguous() and a.shape[1] == k and tuple(w.shape) == (n, k)
            blob += struct.pack("<5Q", a.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else 0, m_dev.data_ptr(),
                                y.data_ptr())
            cap = max(cap, int(a.shape[0]))
        table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(self.device)
        check(self._lib.gigl_linear_grouped(self._ctx, C.c_void_p(table.data_ptr()), len(groups), cap, int(k), int(n), int(act)),
              self._ctx)
        self._keep_groups = table  # (read by the launch: kept until the next call)

    def gather_mean_backward_transposed(self, dout: torch.Tensor, d: int, rowptr: torch.Tensor, rowend: Optional[torch.Tensor],
                                        col: torch.Tensor, n_rows_dev: torch.Tensor, rows_cap: int, n_src_dev: torch.Tensor,
                                        dsrc: torch.Tensor, edges_cap: Optional[int] = None, aggr: str = "mean") -> None:
        """dsrc[j] for j < *n_src_dev = the same gradient, every row WRITTEN once by a gather over the transposed rows
        (gigl_gather_mean_backward_transposed: no zero-fill, no float atomics); "mean" or "sum"; d % 4 == 0"""
        if rowend is None:
            rowend = rowptr[1:]
        assert dout.is_cuda and dout.is_contiguous() and dsrc.is_contiguous() and dout.dtype == torch.float32
        p = lambda t: C.c_void_p(t.data_ptr())
        check(self._lib.gigl_gather_mean_backward_transposed(
            self._ctx, p(dout), d, p(rowptr), p(rowend), p(col), p(n_rows_dev), rows_cap, p(n_src_dev), int(dsrc.shape[0]),
            int(col.numel()) if edges_cap is None else edges_cap, _lib.AGGR[aggr], p(dsrc)), self._ctx)

    def linear(self, a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], m_dev: torch.Tensor,
               m_cap: int, act: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert a.is_cuda and w.is_cuda and a.dtype == torch.float32 and w.dtype == torch.float32
        assert a.is_contiguous() and w.is_contiguous()
        n, k = int(w.shape[0]), int(w.shape[1])
        assert a.shape[1] == k
        if out is None:
            out = torch.empty((m_cap, n), dtype=torch.float32, device=self.device)
        check(self._lib.gigl_linear(self._ctx, C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()),
                                    C.c_void_p(bias.data_ptr()) if bias is not None else None,
                                    C.c_void_p(m_dev.data_ptr()), m_cap, k, n, act, C.c_void_p(out.data_ptr())),
              self._ctx)
        return out

    def linear_weight_grad(self, dy: torch.Tensor, a: torch.Tensor, m_dev: torch.Tensor, relu_y: Optional[torch.Tensor] = None,
                           want_bias: bool = False):
        """(dW [n, k], db [n] | None) = (dy^T a, column sums of dy) over the first *m_dev rows (gigl_linear_weight_grad);
        relu_y: the layer's activated output, masks dy by relu_y > 0"""
        assert dy.is_cuda and a.is_cuda and dy.is_contiguous() and a.is_contiguous() and dy.shape[0] == a.shape[0]
        n, k = int(dy.shape[1]), int(a.shape[1])
        dw = torch.zeros((n, k), dtype=torch.float32, device=self.device)
        db = torch.zeros(n, dtype=torch.float32, device=self.device) if want_bias else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(self._lib.gigl_linear_weight_grad(self._ctx, p(dy), p(a), p(relu_y), p(m_dev), int(dy.shape[0]), n, k, p(dw),
                                                p(db)), self._ctx)
        return dw, db

    # ---- heterogeneous encoders (csrc/hetero.hip) ------------------------------------------------
    def hgt_aggregate(self, q, k, v, heads: int, dim: int, rowptr, col, etype, p_rel, n_dst: int, out) -> None:
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        for t in (q, k, v, out):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
        check(self._lib.gigl_hgt_aggregate(self._ctx, p(q), p(k), p(v), heads, dim, p(rowptr), p(col), p(etype),
                                           p(p_rel), n_dst, p(out)), self._ctx)

    def hgt_aggregate_backward(self, q, k, v, heads: int, dim: int, rowptr, col, etype, p_rel, n_dst: int, out, dout):
        """-> (dq, dk, dv, dp_rel): gigl_hgt_aggregate_backward"""
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        dout = dout.contiguous()
        dq = torch.zeros_like(q)
        dk, dv = torch.zeros_like(k), torch.zeros_like(v)
        dp = torch.zeros_like(p_rel) if p_rel is not None else None
        check(self._lib.gigl_hgt_aggregate_backward(self._ctx, p(q), p(k), p(v), heads, dim, p(rowptr), p(col), p(etype),
                                                    p(p_rel), n_dst, p(out), p(dout), p(dq), p(dk), p(dv), p(dp)),
              self._ctx)
        return dq, dk, dv, dp

    def weighted_aggregate_backward(self, alpha, v, heads: int, dim: int, rowptr, col, n_dst: int, dout):
        """-> (dalpha [E, heads], dv): gigl_weighted_aggregate_backward"""
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        dout = dout.contiguous()
        dalpha, dv = torch.zeros_like(alpha), torch.zeros_like(v)
        check(self._lib.gigl_weighted_aggregate_backward(self._ctx, p(alpha), p(v), heads, dim, p(rowptr), p(col), n_dst,
                                                         p(dout), p(dalpha), p(dv)), self._ctx)
        return dalpha, dv

    def simplehgn_alpha(self, hl, hr, het, hef, src, dst, etype, n_nodes: int, heads: int, slope: float):
        """-> alpha [E, heads] (softmax over the edges sharing a SOURCE node), in the order of src / dst / etype"""
        ne = int(src.numel())
        alpha = torch.empty((ne, heads), dtype=torch.float32, device=self.device)
        scratch = torch.empty(2 * max(n_nodes, 1) * heads, dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(self._lib.gigl_simplehgn_alpha(self._ctx, p(hl), p(hr), p(het), p(hef), p(src), p(dst), p(etype), ne,
                                             n_nodes, heads, float(slope), p(scratch), p(alpha)), self._ctx)
        return alpha

    def weighted_aggregate(self, alpha, v, heads: int, dim: int, rowptr, col, n_dst: int, out) -> None:
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        assert v.is_contiguous() and alpha.is_contiguous() and out.is_contiguous()
        check(self._lib.gigl_weighted_aggregate(self._ctx, p(alpha), p(v), heads, dim, p(rowptr), p(col), n_dst,
                                                p(out)), self._ctx)

    def retrieval_loss(self, scores: torch.Tensor, temperature: Optional[float], cand_prob: Optional[torch.Tensor],
                       query_ids: Optional[torch.Tensor], cand_ids: Optional[torch.Tensor], want_masked: bool = False):
        """-> (loss [] fp32, row_lse [q], masked logits [q, c] | None): gigl_retrieval_loss"""
        assert scores.is_cuda and scores.dtype == torch.float32 and scores.dim() == 2 and scores.stride(1) == 1
        q, c = int(scores.shape[0]), int(scores.shape[1])
        dev = self.device
        lse = torch.empty(q, dtype=torch.float32, device=dev)
        row_loss = torch.empty(q, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        masked = torch.empty((q, c), dtype=torch.float32, device=dev) if want_masked else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(self._lib.gigl_retrieval_loss(self._ctx, p(scores), int(scores.stride(0)), q, c,
                                            float(temperature) if temperature is not None else 0.0, p(cand_prob),
                                            p(query_ids), p(cand_ids), p(masked), p(lse), p(row_loss), p(loss)),
              self._ctx)
        return loss, lse, masked

    def linear_batched(self, a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """a [G, M, K] (rows contiguous; the batch stride is free), w [G, N, K] -> y [M, G, N] with
        y[:, g] = a[g] @ w[g]^T: gigl_linear_batched (one launch for the G products)"""
        assert a.is_cuda and a.dtype == w.dtype == torch.float32 and a.dim() == w.dim() == 3
        g, m, k = (int(v) for v in a.shape)
        n = int(w.shape[1])
        assert w.shape[0] == g and w.shape[2] == k and a.stride(2) == 1 and a.stride(1) == k
        assert w.stride(2) == 1 and w.stride(1) == k
        y = torch.empty((m, g, n), dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(self._lib.gigl_linear_batched(self._ctx, p(a), p(w), None, p(dev_i32(self.device, m)), m, k, n, 0, g,
                                            int(a.stride(0)), int(w.stride(0)), g * n, p(y)), self._ctx)
        return y

    def retrieval_loss_batched(self, scores: torch.Tensor, temperature: Optional[float],
                               cand_prob: Optional[torch.Tensor], query_ids: Optional[torch.Tensor],
                               cand_ids: Optional[torch.Tensor]) -> torch.Tensor:
        """scores [Q, G, C] (as linear_batched writes them), query_ids [G, Q], cand_ids / cand_prob [G, C] -> the G
        losses: gigl_retrieval_loss_batched (forward only)"""
        assert scores.is_cuda and scores.dtype == torch.float32 and scores.dim() == 3 and scores.is_contiguous()
        q, g, c = (int(v) for v in scores.shape)
        for t, n in ((cand_prob, c), (query_ids, q), (cand_ids, c)):
            assert t is None or (t.is_contiguous() and tuple(t.shape) == (g, n))
        assert query_ids is None or query_ids.dtype == torch.int64
        assert cand_ids is None or cand_ids.dtype == torch.int64
        lse = torch.empty((g, q), dtype=torch.float32, device=self.device)
        row_loss = torch.empty((g, q), dtype=torch.float32, device=self.device)
        loss = torch.empty(g, dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(self._lib.gigl_retrieval_loss_batched(self._ctx, p(scores), g * c, c, q, c, g,
                                                    float(temperature) if temperature is not None else 0.0,
                                                    p(cand_prob), p(query_ids), p(cand_ids), p(lse), p(row_loss),
                                                    p(loss)), self._ctx)
        return loss

    def retrieval_loss_backward(self, scores, temperature, cand_prob, query_ids, cand_ids, lse, grad_loss):
        q, c = int(scores.shape[0]), int(scores.shape[1])
        d = torch.empty((q, c), dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(self._lib.gigl_retrieval_loss_backward(self._ctx, p(scores), int(scores.stride(0)), q, c,
                                                     float(temperature) if temperature is not None else 0.0,
                                                     p(cand_prob), p(query_ids), p(cand_ids), p(lse), p(grad_loss),
                                                     p(d)), self._ctx)
        return d

    def project_features(self, w_fused: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[X W_l^T | X W_r^T] over the whole resident feature table for a first-layer fused weight [out, 2*in]
        (gigl_sage_project_features): one fp32 table [n_rows, 2*out] on the device — the input of
        SagePlan.set_projected_input"""
        assert self._feat is not None and w_fused.dim() == 2 and w_fused.shape[1] == 2 * self.feat_dim
        w = w_fused.detach().to(device=self.device, dtype=torch.float32).contiguous()
        n_out = int(w.shape[0])
        n_rows = C.c_int64()
        check(self._lib.gigl_features_device_ptr(self._feat, None, C.byref(n_rows), None, None), self._ctx)
        if out is None:  # (a job re-projects after every weight update: pass the previous table back in)
            out = torch.empty((n_rows.value, 2 * n_out), dtype=torch.float32, device=self.device)
        assert out.is_cuda and out.is_contiguous() and tuple(out.shape) == (n_rows.value, 2 * n_out)
        check(self._lib.gigl_sage_project_features(self._ctx, self._feat, C.c_void_p(w.data_ptr()), n_out,
                                                   C.c_void_p(out.data_ptr())), self._ctx)
        return out

    # ---- one-call batch pipeline -----------------------------------------------------------
    def make_sage_plan(self, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]], b: int,
                       fanouts: Sequence[int], act_last: bool = False, groups: int = 1, aggr: str = "mean") -> "SagePlan":
        """weights[l]: fused fp32 [out_l, 2*in_l] = cat(lin_l.weight, lin_r.weight, dim=1) on this device.
        groups > 1: one call takes groups*b roots = `groups` independent batches of b roots each.
        aggr: "mean" | "sum" | "max" (PyG SAGEConv aggr)."""
        plan = SagePlan(self, weights, biases, b, fanouts, act_last, groups)
        if aggr != "mean":
            check(self._lib.gigl_sage_plan_set_aggr(plan._plan, _lib.AGGR[aggr]), self._ctx)
        return plan


from collections import OrderedDict

_DEV_I32: "OrderedDict" = OrderedDict()
_DEV_I32_MAX = 256


def dev_i32(device, value: int) -> torch.Tensor:
    """a cached one-element int32 device tensor (row counts handed to the library by pointer): created once per
    (device index, value) — a fresh torch.tensor([...], device=...) per call is a pageable host-to-device copy, which
    synchronises and cannot be captured in a graph.  A small LRU: callers pass per-batch sizes (numbers of candidates),
    so the set of values is unbounded over a job; an evicted tensor stays alive while a launch still holds it."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else -1)
    key = (dev.type, idx, int(value))
    t = _DEV_I32.get(key)
    if t is None:
        t = torch.tensor([int(value)], dtype=torch.int32, device=device)
        _DEV_I32[key] = t
        if len(_DEV_I32) > _DEV_I32_MAX:
            _DEV_I32.popitem(last=False)
    else:
        _DEV_I32.move_to_end(key)
    return t


class SagePlan:
    """sample -> union -> GraphSAGE forward -> one row per root, enqueued by ONE library call
    (include/gigl_hip.h `gigl_sage_plan_*`)."""

    def __init__(self, eng: HipEngine, weights, biases, b: int, fanouts, act_last: bool, groups: int = 1):
        assert eng._graph is not None and eng._feat is not None, "load the graph and the features first"
        L = len(fanouts)
        assert len(weights) == L and groups >= 1
        self.group_roots, self.groups = int(b), int(groups)
        self.eng, self.b, self.fanouts = eng, int(b) * int(groups), [int(f) for f in fanouts]
        self._lib = eng._lib
        self.dims = [int(weights[0].shape[1]) // 2] + [int(w.shape[0]) for w in weights]
        self._keep = None
        self._plan = C.c_void_p()
        w_arr, b_arr = self._ptr_arrays(weights, biases)
        fo = (C.c_int32 * L)(*self.fanouts)
        dims = (C.c_int32 * (L + 1))(*self.dims)
        check(self._lib.gigl_sage_plan_create(eng._ctx, eng._graph, eng._feat, self.b, fo, L, dims, w_arr, b_arr,
                                              1 if act_last else 0, C.byref(self._plan)), eng._ctx)
        if self.groups > 1:
            check(self._lib.gigl_sage_plan_set_groups(self._plan, self.group_roots), eng._ctx)
        if not hasattr(eng, "_plans"):
            eng._plans = []
        eng._plans.append(self)

    def _ptr_arrays(self, weights, biases):
        L = len(weights)
        ws = [w.detach().to(device=self.eng.device, dtype=torch.float32).contiguous() for w in weights]
        bs = [None if x is None else x.detach().to(device=self.eng.device, dtype=torch.float32).contiguous()
              for x in biases]
        self._keep = (ws, bs)  # the plan borrows these device buffers
        w_arr = (C.c_void_p * L)(*[w.data_ptr() for w in ws])
        b_arr = (C.c_void_p * L)(*[(x.data_ptr() if x is not None else None) for x in bs])
        return w_arr, b_arr

    def set_weights(self, weights, biases) -> None:
        w_arr, b_arr = self._ptr_arrays(weights, biases)
        check(self._lib.gigl_sage_plan_set_weights(self._plan, w_arr, b_arr), self.eng._ctx)

    def set_projected_input(self, proj: Optional[torch.Tensor]) -> None:
        """first layer over PROJECTED rows (HipEngine.project_features of this plan's first-layer weight): one reduction
        + the self row + bias, no per-batch projection (gigl_sage_plan_set_projected_input); None switches back.
        The table belongs to the weights it was computed from: recompute it after a weight update."""
        if proj is None:
            check(self._lib.gigl_sage_plan_set_projected_input(self._plan, None), self.eng._ctx)
            self._proj = None
            return
        assert proj.is_cuda and proj.dtype == torch.float32 and proj.is_contiguous() and proj.shape[1] == 2 * self.dims[1]
        check(self._lib.gigl_sage_plan_set_projected_input(self._plan, C.c_void_p(proj.data_ptr())), self.eng._ctx)
        self._proj = proj  # borrowed by the plan

    def half_split(self) -> bool:
        """the first projection runs over two fp16 planes per operand (gigl_sage_plan_half_split)"""
        return bool(self._lib.gigl_sage_plan_half_split(self._plan))

    def fused_layers(self) -> bool:
        """both SAGE layers' projections run in one kernel and the last layer is one reduction over [W_l h | W_r h] rows
        (gigl_sage_plan_fused_layers)"""
        return bool(self._lib.gigl_sage_plan_fused_layers(self._plan))

    def fused_planes(self) -> int:
        """partial planes of p rows the fused projection writes per node (0: layers apart; 1: whole rows; 2: K-split)"""
        return int(self._lib.gigl_sage_plan_fused_layers(self._plan))

    def set_graph_stream(self, stream: Optional[torch.cuda.Stream]) -> None:
        """issue the graph part (sample + union) of every later call on `stream` — typically
        torch.cuda.Stream(priority=-1), a higher priority than the engine's stream — with the layers waiting on an event
        (gigl_sage_plan_set_graph_stream); None: back to one stream"""
        self._graph_stream = stream  # (kept alive with the plan)
        check(self._lib.gigl_sage_plan_set_graph_stream(self._plan, C.c_void_p(stream.cuda_stream) if stream is not None else None,
                                                        1 if stream is not None else 0), self.eng._ctx)

    def overflow_add(self, acc: torch.Tensor) -> None:
        """acc (int32 [1], device) += 1 when the batch set run last failed (its rows are NaN): gigl_sage_plan_overflow_add,
        no synchronisation"""
        assert acc.is_cuda and acc.dtype == torch.int32 and acc.numel() >= 1
        check(self._lib.gigl_sage_plan_overflow_add(self._plan, C.c_void_p(acc.data_ptr())), self.eng._ctx)

    def use_graph(self, on: bool = True) -> None:
        """replay the batch as one hipGraph launch (captured on the next run)"""
        check(self._lib.gigl_sage_plan_use_graph(self._plan, 1 if on else 0), self.eng._ctx)

    def flush_profile(self) -> None:
        check(self._lib.gigl_sage_plan_flush_profile(self._plan), self.eng._ctx)

    def run(self, roots: torch.Tensor, out: Optional[torch.Tensor] = None, sampling_seed: int = 42,
            mode: int = MODE_SPARK_HASH) -> torch.Tensor:
        """roots: int32 device tensor [b] (uint32 ids); returns [b, out_dim] (row i <-> roots[i])"""
        assert roots.is_cuda and roots.dtype == torch.int32 and roots.numel() == self.b and roots.is_contiguous()
        if out is None:
            out = torch.empty((self.b, self.dims[-1]), dtype=torch.float32, device=self.eng.device)
        check(self._lib.gigl_sage_plan_run(self._plan, C.c_void_p(roots.data_ptr()), sampling_seed, mode,
                                           C.c_void_p(out.data_ptr())), self.eng._ctx)
        return out

    def stats(self, roots: torch.Tensor, acc: torch.Tensor) -> None:
        """add the exact work counts of the batch set run last (with `roots`) to acc (int64 [STATS_LEN], device)"""
        assert acc.is_cuda and acc.dtype == torch.int64 and acc.numel() >= _lib.STATS_LEN
        check(self._lib.gigl_sage_plan_stats(self._plan, C.c_void_p(roots.data_ptr()), C.c_void_p(acc.data_ptr())),
              self.eng._ctx)

    def _d2h(self, ptr, n, dtype):
        a = np.empty(n, dtype=dtype)
        if n:
            check(self._lib.gigl_memcpy(self.eng._ctx, C.c_void_p(a.ctypes.data), LOC_HOST, C.c_void_p(ptr), LOC_DEVICE,
                                        a.nbytes), self.eng._ctx)
        return a

    def last_batch_to_host(self):
        """host copies of the last batch's tree and union graph:
        dict(nbr=[...], cnt=[...], meta, nodes, rowptr, rowend, col, root_local)"""
        t, u = GiglTree(), GiglUnion()
        check(self._lib.gigl_sage_plan_buffers(self._plan, C.byref(t), C.byref(u)), self.eng._ctx)
        nbr, cnt, parents = [], [], self.b
        for k, f in enumerate(self.fanouts):
            cnt.append(self._d2h(t.cnt[k], parents, np.int32))
            parents *= f
            nbr.append(self._d2h(t.nbr[k], parents, np.uint32))
        meta = self._d2h(u.meta, GIGL_META_LEN, np.int32)
        nn = int(meta[0])
        return dict(nbr=nbr, cnt=cnt, meta=meta, nodes=self._d2h(u.nodes, nn, np.uint32),
                    rowptr=self._d2h(u.rowptr, nn + 1, np.int32), rowend=self._d2h(u.rowend, nn + 1, np.int32),
                    col=self._d2h(u.col, int(u.cap_edges), np.int32), root_local=self._d2h(u.root_local, self.b, np.int32))

    def close(self) -> None:
        if getattr(self, "_plan", None):
            self._lib.gigl_sage_plan_destroy(self._plan)
            self._plan = None
            if self in getattr(self.eng, "_plans", []):
                self.eng._plans.remove(self)


class SageTrainPlan:
    """one TRAINING step per call in the library (include/gigl_hip.h `gigl_sage_train_plan_*`): sample -> union graph ->
    GraphSAGE forward -> cross-entropy on the roots -> backward -> Adam, replayed as one hipGraph per step.  The plan
    trains FUSED weights [W_l | W_r] (and biases) held here as torch tensors and updated in place by every step;
    `load(model)` / `store(model)` move them from / to a models.GraphSAGE (Adam is element-wise: training the fused
    matrix is training lin_l.weight and lin_r.weight)."""

    def __init__(self, eng: HipEngine, model, b: int, fanouts, lr: float = 0.01, weight_decay: float = 5e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8):
        assert eng._graph is not None and eng._feat is not None, "load the graph and the features first"
        L = len(fanouts)
        assert model.num_layers == L, "one hop per layer"
        if not (model._plain and model.aggr == "mean" and not model.should_l2_normalize_embedding_layer_output
                and model.feats_interaction is None and model.feature_embedding_layer is None
                and all(c.lin_r is not None for c in model.conv_layers)):
            raise NotImplementedError("the training plan runs plain mean-GraphSAGE layers (conv -> relu)")
        self.eng, self.b, self.fanouts = eng, int(b), [int(f) for f in fanouts]
        self._lib = eng._lib
        self.w, self.bias = [], []
        self.load(model)
        self.dims = [int(self.w[0].shape[1]) // 2] + [int(w.shape[0]) for w in self.w]
        w_arr = (C.c_void_p * L)(*[w.data_ptr() for w in self.w])
        b_arr = (C.c_void_p * L)(*[(x.data_ptr() if x is not None else None) for x in self.bias])
        fo = (C.c_int32 * L)(*self.fanouts)
        dims = (C.c_int32 * (L + 1))(*self.dims)
        act_last = 1 if model.activation_after_last_conv else 0

        def create():
            h = C.c_void_p()
            check(self._lib.gigl_sage_train_plan_create(eng._ctx, eng._graph, eng._feat, self.b, fo, L, dims, w_arr, b_arr,
                                                        act_last, float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                        float(weight_decay), C.byref(h)), eng._ctx)
            return h
        self._create, self._adopt, self._destroy = create, self._lib.gigl_sage_train_plan_adopt, self._lib.gigl_sage_train_plan_destroy
        self.wide = False
        self._plan = create()
        self.loss = torch.zeros(1, dtype=torch.float32, device=eng.device)  # the last step's loss (device scalar)

    def load(self, model) -> None:
        """(re)read the model's parameters into the fused buffers the plan trains (in place when they exist)"""
        ws = [c.fused_weight().detach().to(device=self.eng.device, dtype=torch.float32).contiguous() for c in model.conv_layers]
        bs = [None if c.lin_l.bias is None else c.lin_l.bias.detach().to(device=self.eng.device, dtype=torch.float32)
              .clone().contiguous() for c in model.conv_layers]
        if self.w:
            for dst, src in zip(self.w, ws):
                dst.copy_(src)
            for dst, src in zip(self.bias, bs):
                if dst is not None:
                    dst.copy_(src)
        else:
            self.w, self.bias = ws, bs

    def store(self, model) -> None:
        """write the trained parameters back into the model (lin_l.weight | lin_r.weight halves of the fused matrix)"""
        with torch.no_grad():
            for c, w, b in zip(model.conv_layers, self.w, self.bias):
                d = c.in_channels
                c.lin_l.weight.copy_(w[:, :d])
                c.lin_r.weight.copy_(w[:, d:])
                if b is not None:
                    c.lin_l.bias.copy_(b)

    def grow(self) -> None:
        """re-create the plan with workspaces for the WORST batch (gigl_ctx_set_wide_workspaces: rows for every node of the
        tree, the generic union build) and hand it the optimiser state of the plan it replaces — the weights are this
        object's tensors, shared by construction.  What step_checked does the first time a batch overflows the regular
        workspace (roots that are each other's sampled neighbours: common on graphs of a few thousand nodes)."""
        if self.wide:
            return
        check(self._lib.gigl_ctx_set_wide_workspaces(self.eng._ctx, 1), self.eng._ctx)
        try:
            new = self._create()
        finally:
            self._lib.gigl_ctx_set_wide_workspaces(self.eng._ctx, 0)
        check(self._adopt(new, self._plan), self.eng._ctx)
        self._destroy(self._plan)
        self._plan, self.wide = new, True
        self._next = None  # (a batch announced to the old plan's workspaces is sampled again)

    def step_checked(self, *args, **kwargs) -> float:
        """step(), its loss read back (synchronises): a batch that did not fit the plan's workspace — NaN loss, nothing
        trained, Adam's counter unmoved — is REDONE after the plan has grown (grow), so the training run sees every batch
        the reference's collate would have built (rooted_node_neighborhood_data_loader.py:78-158 has no such bound).  A
        NaN from a plan that is wide already is the loss's own."""
        loss = float(self.step(*args, **kwargs)[0])
        if loss != loss and not self.wide:
            self.grow()
            self.overflow_redone = getattr(self, "overflow_redone", 0) + 1
            loss = float(self.step(*args, **kwargs)[0])
        return loss

    def _padded(self, roots: torch.Tensor) -> torch.Tensor:
        k = int(roots.numel())
        assert roots.is_cuda and roots.dtype == torch.int32 and 0 < k <= self.b
        if k < self.b:
            roots = torch.cat([roots, roots[:1].expand(self.b - k)])
        return roots.contiguous()

    def moments(self) -> dict:
        """Adam's moments as torch.optim.Adam names them, keyed like the model's state dict: {"conv_layers.l.lin_l.weight":
        (exp_avg, exp_avg_sq), ...lin_r.weight, ...lin_l.bias} (gigl_sage_train_plan_moments; the fused matrix split back)"""
        out = {}
        for l, (w, b) in enumerate(zip(self.w, self.bias)):
            mw, vw = torch.empty_like(w), torch.empty_like(w)
            mb = vb = None
            if b is not None:
                mb, vb = torch.empty_like(b), torch.empty_like(b)
            p_ = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
            check(self._lib.gigl_sage_train_plan_moments(self._plan, l, p_(mw), p_(vw), p_(mb), p_(vb)), self.eng._ctx)
            d = int(w.shape[1]) // 2
            out[f"conv_layers.{l}.lin_l.weight"] = (mw[:, :d], vw[:, :d])
            out[f"conv_layers.{l}.lin_r.weight"] = (mw[:, d:], vw[:, d:])
            if b is not None:
                out[f"conv_layers.{l}.lin_l.bias"] = (mb, vb)
        return out

    def resume(self) -> None:
        """clear the device-side halt a failed batch left (gigl_sage_train_plan_resume): later steps train again"""
        check(self._lib.gigl_sage_train_plan_resume(self._plan), self.eng._ctx)

    def run_steps(self, n_steps: int, step_args) -> torch.Tensor:
        """steps 0 .. n_steps-1 issued back to back WITHOUT a host read (step_args(i) -> the kwargs of step i), their losses
        collected on the device -> float32 [n_steps].  A batch that does not fit the plan's workspace halts the plan on the
        device (nothing behind it is applied); the loop then grows the plan (grow) and redoes the steps from that batch on.
        A batch that fails in a wide plan (a label outside the output width) is skipped, as before."""
        losses = torch.zeros(max(n_steps, 1), dtype=torch.float32, device=self.eng.device)
        start = 0
        while start < n_steps:
            for i in range(start, n_steps):
                self.step(**step_args(i), loss_out=losses[i:i + 1])
            bad = torch.isnan(losses[start:n_steps]).nonzero()  # (synchronises: once per pass, not per step)
            if bad.numel() == 0:
                break
            j = start + int(bad[0])
            if not self.wide:
                self.grow()
                self.overflow_redone = getattr(self, "overflow_redone", 0) + 1
                start = j
            else:
                self.resume()
                start = j + 1
        return losses[:n_steps]

    def step(self, roots: torch.Tensor, labels: torch.Tensor, sampling_seed: int = 42, mode: int = MODE_SPARK_HASH,
             next_roots: Optional[torch.Tensor] = None, next_roots2: Optional[torch.Tensor] = None,
             loss_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """one optimiser step on the batch: roots int32 device [k <= b] (uint32 ids), labels int64 device [k]; a short
        batch is padded with its first root (a repeated root adds nothing to the union graph) and masked out of the loss.
        next_roots: the roots of the batch the NEXT call will be given — its sampling and union graph then run on the
        plan's own stream while this step's forward / backward run (roots_next of gigl_sage_train_plan_step);
        next_roots2: the batch after that one (two graph parts in flight: gigl_sage_train_plan_step2).
        Returns the loss (a device scalar owned by the plan, overwritten by the next step)."""
        k = int(roots.numel())
        assert labels.is_cuda and labels.dtype == torch.int64 and labels.numel() == k
        roots, labels = self._padded(roots), labels.contiguous()
        nxt = self._padded(next_roots) if next_roots is not None and next_roots.numel() else None
        nxt2 = self._padded(next_roots2) if nxt is not None and next_roots2 is not None and next_roots2.numel() else None
        # (read asynchronously by the plan's own streams: kept alive over the next three steps)
        self._keep = (tuple(getattr(self, "_keep", ()))[-2:]) + ((roots, nxt, nxt2),)
        p_ = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        out = self.loss if loss_out is None else loss_out  # (loss_out: a float32 device slot of the caller's)
        check(self._lib.gigl_sage_train_plan_step2(self._plan, p_(roots), p_(labels), k, p_(nxt), p_(nxt2),
                                                   int(sampling_seed), int(mode), p_(out)), self.eng._ctx)
        return out

    def close(self) -> None:
        if getattr(self, "_plan", None):
            self._lib.gigl_sage_train_plan_destroy(self._plan)
            self._plan = None
        self._create = None  # (drops the argument arrays it holds)


class NablpTrainPlan:
    """one LINK-PREDICTION training step per call in the library (include/gigl_hip.h `gigl_nablp_train_plan_*`): sample +
    union of the main batch (anchors with their positives) and of the random-negative batch -> two GraphSAGE forwards over
    the shared weights -> inner-product scores -> retrieval loss -> backward of both encodes -> Adam, replayed as one
    hipGraph per step.  Like SageTrainPlan it trains FUSED weights [W_l | W_r] held here as torch tensors (`load` / `store`
    move them from / to a models.GraphSAGE)."""

    def __init__(self, eng: HipEngine, model, b_anchors: int, num_positives: int, n_random_negatives: int, fanouts,
                 temperature: float = 0.07, remove_accidental_hits: bool = True, lr: float = 5e-3,
                 weight_decay: float = 1e-6, betas=(0.9, 0.999), eps: float = 1e-8):
        assert eng._graph is not None and eng._feat is not None, "load the graph and the features first"
        L = len(fanouts)
        assert model.num_layers == L, "one hop per layer"
        if not (model._plain and model.aggr == "mean" and model.feats_interaction is None
                and model.feature_embedding_layer is None and all(c.lin_r is not None for c in model.conv_layers)):
            raise NotImplementedError("the link-prediction training plan runs plain mean-GraphSAGE layers (conv -> relu)")
        self.eng, self.b, self.P, self.n_rn = eng, int(b_anchors), int(num_positives), int(n_random_negatives)
        self.fanouts = [int(f) for f in fanouts]
        self._lib = eng._lib
        self.w, self.bias = [], []
        SageTrainPlan.load(self, model)
        self.dims = [int(self.w[0].shape[1]) // 2] + [int(w.shape[0]) for w in self.w]
        w_arr = (C.c_void_p * L)(*[w.data_ptr() for w in self.w])
        b_arr = (C.c_void_p * L)(*[(x.data_ptr() if x is not None else None) for x in self.bias])
        fo = (C.c_int32 * L)(*self.fanouts)
        dims = (C.c_int32 * (L + 1))(*self.dims)
        flags = (1 if model.activation_after_last_conv else 0, 1 if model.should_l2_normalize_embedding_layer_output else 0)

        def create():
            h = C.c_void_p()
            check(self._lib.gigl_nablp_train_plan_create(
                eng._ctx, eng._graph, eng._feat, self.b, self.P, self.n_rn, fo, L, dims, w_arr, b_arr, flags[0], flags[1],
                float(temperature), 1 if remove_accidental_hits else 0, float(lr), float(betas[0]), float(betas[1]), float(eps),
                float(weight_decay), C.byref(h)), eng._ctx)
            return h
        self._create, self._adopt, self._destroy = create, self._lib.gigl_nablp_train_plan_adopt, self._lib.gigl_nablp_train_plan_destroy
        self.wide = False
        self._plan = create()
        self.loss = torch.zeros(2, dtype=torch.float32, device=eng.device)  # {loss, query rows} of the last step

    load = SageTrainPlan.load
    store = SageTrainPlan.store
    grow = SageTrainPlan.grow
    step_checked = SageTrainPlan.step_checked

    def _padded(self, main_roots: torch.Tensor, pos_cnt: Optional[torch.Tensor], rn_roots: torch.Tensor):
        T = 1 + self.P
        assert main_roots.is_cuda and main_roots.dtype == torch.int32 and main_roots.numel() % T == 0
        na = main_roots.numel() // T
        assert 0 < na <= self.b and (pos_cnt is None or pos_cnt.numel() == na) and rn_roots.numel() <= self.n_rn
        if na < self.b:
            main_roots = torch.cat([main_roots, torch.full(((self.b - na) * T,), -1, dtype=torch.int32, device=main_roots.device)])
            if pos_cnt is not None:
                pos_cnt = torch.cat([pos_cnt, torch.zeros(self.b - na, dtype=torch.int32, device=pos_cnt.device)])
        if rn_roots.numel() < self.n_rn:
            rn_roots = torch.cat([rn_roots, torch.full((self.n_rn - rn_roots.numel(),), -1, dtype=torch.int32,
                                                       device=main_roots.device)])
        return (main_roots.contiguous(), None if pos_cnt is None else pos_cnt.to(torch.int32).contiguous(), rn_roots.contiguous())

    def step(self, main_roots: torch.Tensor, pos_cnt: torch.Tensor, rn_roots: torch.Tensor, sampling_seed: int = 42,
             mode: int = MODE_SPARK_HASH, next_roots=None) -> torch.Tensor:
        """main_roots int32 device [n_anchors * (1 + P)] anchor-major (anchor, its positive slots), pos_cnt int32 device
        [n_anchors], rn_roots int32 device [<= n_random_negatives]; short batches are padded here (absent anchors /
        negatives = 0xFFFFFFFF).  next_roots = (main_roots, rn_roots) of the NEXT step: its sampling and union build then
        run on a side stream beside this step's layers (gigl_nablp_train_plan_step2); the next call recognises them by
        value.  Returns {loss, query rows} (device, owned by the plan)."""
        pre, self._next = getattr(self, "_next", None), None
        key = (main_roots.data_ptr(), rn_roots.data_ptr(), main_roots.numel(), rn_roots.numel())
        main_roots, pos_cnt, rn_roots = self._padded(main_roots, pos_cnt, rn_roots)
        if pre is not None and pre["key"] == key:  # the (padded) buffers the previous call announced to the plan
            main_roots, rn_roots = pre["main"], pre["rn"]
        nxt_m = nxt_r = None
        if next_roots is not None:
            nxt_m, _, nxt_r = self._padded(next_roots[0], None, next_roots[1])
            # (the caller's tensors are kept alive: their addresses identify the batch at the next call)
            self._next = dict(main=nxt_m, rn=nxt_r, keep=tuple(next_roots),
                              key=(next_roots[0].data_ptr(), next_roots[1].data_ptr(), next_roots[0].numel(), next_roots[1].numel()))
        self._keep = (main_roots, pos_cnt, rn_roots, nxt_m, nxt_r)
        p_ = lambda t: C.c_void_p(t.data_ptr())
        check(self._lib.gigl_nablp_train_plan_step2(
            self._plan, p_(main_roots), p_(pos_cnt), p_(rn_roots) if self.n_rn else None,
            p_(nxt_m) if nxt_m is not None else None, p_(nxt_r) if (nxt_r is not None and self.n_rn) else None,
            int(sampling_seed), int(mode), p_(self.loss)), self.eng._ctx)
        return self.loss

    def grads(self, layer: int):
        """(d loss / d [W_l | W_r], d loss / d bias) of `layer` from the LAST step (gigl_nablp_train_plan_grads)"""
        gw = torch.empty_like(self.w[layer])
        gb = torch.empty_like(self.bias[layer]) if self.bias[layer] is not None else None
        check(self._lib.gigl_nablp_train_plan_grads(self._plan, int(layer), C.c_void_p(gw.data_ptr()),
                                                    C.c_void_p(gb.data_ptr()) if gb is not None else None), self.eng._ctx)
        return gw, gb

    def moments(self) -> dict:
        """as SageTrainPlan.moments (gigl_nablp_train_plan_moments)"""
        out = {}
        for l, (w, b) in enumerate(zip(self.w, self.bias)):
            mw, vw = torch.empty_like(w), torch.empty_like(w)
            check(self._lib.gigl_nablp_train_plan_moments(self._plan, 2 * l, C.c_void_p(mw.data_ptr()), C.c_void_p(vw.data_ptr())),
                  self.eng._ctx)
            d = int(w.shape[1]) // 2
            out[f"conv_layers.{l}.lin_l.weight"] = (mw[:, :d], vw[:, :d])
            out[f"conv_layers.{l}.lin_r.weight"] = (mw[:, d:], vw[:, d:])
            if b is not None:
                mb, vb = torch.empty_like(b), torch.empty_like(b)
                check(self._lib.gigl_nablp_train_plan_moments(self._plan, 2 * l + 1, C.c_void_p(mb.data_ptr()),
                                                              C.c_void_p(vb.data_ptr())), self.eng._ctx)
                out[f"conv_layers.{l}.lin_l.bias"] = (mb, vb)
        return out

    def close(self) -> None:
        if getattr(self, "_plan", None):
            self._lib.gigl_nablp_train_plan_destroy(self._plan)
            self._plan = None
        self._create = None


class GatNablpTrainPlan(NablpTrainPlan):
    """the link-prediction training step with the two-layer GAT encoder (gigl_gat_nablp_train_plan_create; step / loss /
    prefetch are NablpTrainPlan's).  Trains copies of the encoder's parameters held here as torch tensors (`load` / `store`
    move them from / to a models_attn.GAT)."""

    def __init__(self, eng: HipEngine, model, b_anchors: int, num_positives: int, n_random_negatives: int, fanouts,
                 temperature: float = 0.07, remove_accidental_hits: bool = True, lr: float = 5e-3,
                 weight_decay: float = 1e-6, betas=(0.9, 0.999), eps: float = 1e-8):
        assert eng._graph is not None and eng._feat is not None, "load the graph and the features first"
        if not self.applies(model, eng.feat_dim) or len(fanouts) != 2:
            raise NotImplementedError("the GAT link-prediction plan runs two plain GATConv layers (concatenated heads, one "
                                      "head in the second, no edge features) whose input rows are wider than the first "
                                      "layer's output")
        self.eng, self.b, self.P, self.n_rn = eng, int(b_anchors), int(num_positives), int(n_random_negatives)
        self.fanouts = [int(f) for f in fanouts]
        self._lib = eng._lib
        self.load(model)
        c0, c1 = model.conv_layers
        self.heads, self.channels = [int(c0.heads), 1], [int(c0.out_channels), int(c1.out_channels)]
        arr = lambda ts: (C.c_void_p * 2)(*[(t.data_ptr() if t is not None else None) for t in ts])
        slope, norm = float(c0.negative_slope), 1 if model.should_l2_normalize_embedding_layer_output else 0

        def create():
            h = C.c_void_p()
            check(self._lib.gigl_gat_nablp_train_plan_create(
                eng._ctx, eng._graph, eng._feat, self.b, self.P, self.n_rn, (C.c_int32 * 2)(*self.fanouts), 2,
                (C.c_int32 * 2)(*self.heads), (C.c_int32 * 2)(*self.channels), arr(self.w), arr(self.att_src), arr(self.att_dst),
                arr(self.bias), slope, norm, float(temperature), 1 if remove_accidental_hits else 0, float(lr), float(betas[0]),
                float(betas[1]), float(eps), float(weight_decay), C.byref(h)), eng._ctx)
            return h
        self._create, self._adopt, self._destroy = create, self._lib.gigl_nablp_train_plan_adopt, self._lib.gigl_nablp_train_plan_destroy
        self.wide = False
        self._plan = create()
        self.loss = torch.zeros(2, dtype=torch.float32, device=eng.device)

    @staticmethod
    def applies(model, feat_dim: int) -> bool:
        from .models_attn import GAT, GATConv
        if type(model) is not GAT or model.num_layers != 2 or model.edge_dim is not None or model.activation_after_last_conv:
            return False
        c0, c1 = model.conv_layers
        if type(c0) is not GATConv or type(c1) is not GATConv or not c0.concat or c1.heads != 1:
            return False
        d = int(c0.in_channels)
        return (d == int(feat_dim) and d % 4 == 0 and d <= 1024 and c0.heads in (1, 2, 4) and d > c0.heads * c0.out_channels
                and c0.heads * c0.out_channels <= 1024 and c0.out_channels % 4 == 0 and c1.out_channels % 4 == 0
                and c1.out_channels <= 512 and c0.negative_slope == c1.negative_slope)

    def load(self, model) -> None:
        dev = self.eng.device
        f = lambda t: None if t is None else t.detach().to(dev, torch.float32).reshape(-1).contiguous().clone()
        self.w = [c.lin.weight.detach().to(dev, torch.float32).contiguous().clone() for c in model.conv_layers]
        self.att_src = [f(c.att_src) for c in model.conv_layers]
        self.att_dst = [f(c.att_dst) for c in model.conv_layers]
        self.bias = [f(c.bias) for c in model.conv_layers]
        if getattr(self, "_plan", None):
            raise RuntimeError("load() before the plan exists (its parameter pointers are baked into the plan)")

    def store(self, model) -> None:
        with torch.no_grad():
            for l, c in enumerate(model.conv_layers):
                c.lin.weight.copy_(self.w[l])
                c.att_src.copy_(self.att_src[l].view_as(c.att_src))
                c.att_dst.copy_(self.att_dst[l].view_as(c.att_dst))
                if c.bias is not None:
                    c.bias.copy_(self.bias[l])

    def grads(self, layer: int):
        """(d W, d att_src, d att_dst, d bias) of `layer` from the LAST step"""
        outs = [torch.empty_like(self.w[layer]), torch.empty_like(self.att_src[layer]), torch.empty_like(self.att_dst[layer]),
                torch.empty_like(self.bias[layer]) if self.bias[layer] is not None else None]
        check(self._lib.gigl_gat_nablp_train_plan_grads(self._plan, int(layer),
                                                        *[(C.c_void_p(t.data_ptr()) if t is not None else None) for t in outs]),
              self.eng._ctx)
        return tuple(outs)


    def moments(self) -> dict:
        """{"conv_layers.l.lin.weight" | att_src | att_dst | bias: (exp_avg, exp_avg_sq)} (gigl_nablp_train_plan_moments)"""
        out = {}
        for l in range(2):
            for k, (name, ts) in enumerate((("lin.weight", self.w), ("att_src", self.att_src), ("att_dst", self.att_dst),
                                            ("bias", self.bias))):
                if ts[l] is None:
                    continue
                m, v = torch.empty_like(ts[l]), torch.empty_like(ts[l])
                check(self._lib.gigl_nablp_train_plan_moments(self._plan, 4 * l + k, C.c_void_p(m.data_ptr()),
                                                              C.c_void_p(v.data_ptr())), self.eng._ctx)
                out[f"conv_layers.{l}.{name}"] = (m, v)
        return out


class GatPlan(SagePlan):
    """sample -> union -> GAT forward -> one row per root, enqueued by ONE library call (gigl_gat_plan_create; the
    handle is a gigl_sage_plan: run / use_graph / stats / last_batch_to_host are SagePlan's)"""

    def __init__(self, eng: HipEngine, weights, att_src, att_dst, biases, heads, channels, b: int, fanouts,
                 negative_slope: float = 0.2, act_last: bool = False, groups: int = 1):
        assert eng._graph is not None and eng._feat is not None, "load the graph and the features first"
        L = len(fanouts)
        assert len(weights) == len(att_src) == len(att_dst) == len(heads) == len(channels) == L and groups >= 1
        self.group_roots, self.groups = int(b), int(groups)
        self.eng, self.b, self.fanouts = eng, int(b) * int(groups), [int(f) for f in fanouts]
        self._lib = eng._lib
        self.dims = [int(weights[0].shape[1])] + [int(h) * int(c) for h, c in zip(heads, channels)]
        self._keep = None
        self._plan = C.c_void_p()
        arrs = self._gat_ptr_arrays(weights, att_src, att_dst, biases)
        fo = (C.c_int32 * L)(*self.fanouts)
        hd, ch = (C.c_int32 * L)(*[int(h) for h in heads]), (C.c_int32 * L)(*[int(c) for c in channels])
        check(self._lib.gigl_gat_plan_create(eng._ctx, eng._graph, eng._feat, self.b, fo, L, hd, ch, *arrs,
                                             float(negative_slope), 1 if act_last else 0, C.byref(self._plan)), eng._ctx)
        if self.groups > 1:
            check(self._lib.gigl_sage_plan_set_groups(self._plan, self.group_roots), eng._ctx)
        if not hasattr(eng, "_plans"):
            eng._plans = []
        eng._plans.append(self)

    def _gat_ptr_arrays(self, weights, att_src, att_dst, biases):
        L = len(weights)
        dev = lambda t: t.detach().to(device=self.eng.device, dtype=torch.float32).reshape(t.shape).contiguous()
        ws, a_s, a_d = [dev(w) for w in weights], [dev(a.reshape(-1)) for a in att_src], [dev(a.reshape(-1)) for a in att_dst]
        bs = [None if x is None else dev(x) for x in biases]
        self._keep = (ws, a_s, a_d, bs)  # the plan borrows these device buffers
        arr = lambda ts: (C.c_void_p * L)(*[(t.data_ptr() if t is not None else None) for t in ts])
        return arr(ws), arr(a_s), arr(a_d), arr(bs)

    def set_weights(self, weights, att_src, att_dst, biases) -> None:
        check(self._lib.gigl_gat_plan_set_weights(self._plan, *self._gat_ptr_arrays(weights, att_src, att_dst, biases)),
              self.eng._ctx)
