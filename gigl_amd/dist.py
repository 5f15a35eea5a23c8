"""Multi-GPU path: hash-partitioned graph, per-hop frontier exchange, feature pull.

One process per GPU (`torch.distributed`; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  Where the reference shards:
  node partition   owner(v) = v % world_size
                   (python/gigl/distributed/dist_link_prediction_data_partitioner.py:692-695); edges follow the
                   DESTINATION node (in-edge sampling), i.e. rank r holds the CSC rows of the nodes it owns
  seeds            each rank takes its own root batches
                   (python/gigl/distributed/distributed_neighborloader.py:195-216)
  remote sampling  GLT RPC per batch (distributed_neighborloader.py:182-192)  -> here ONE all_to_all(v) pair per
                   hop: requests (node id, K) out to the owners, f sampled ids per request back.  The pattern is
                   a full mesh, so all 7 xGMI links of a GPU carry traffic at once (no ring).
  feature pull     after the batch union graph is built, the UNIQUE node ids go to their owners and the rows
                   come back (dedup before the pull: unique << sampled)
The owner-side expansion is a callable: `HipEngine.expand_frontier` in production (gigl_expand_frontier on the
rank's shard); tests on CPU plug in the oracle.  The exchange itself is backend-agnostic torch.distributed code.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

INVALID = 0xFFFFFFFF


def _check(rc, ctx=None):
    from ._lib import check
    check(rc, ctx)


def node_owner(ids: torch.Tensor, world: int) -> torch.Tensor:
    return ids % world


def partition_csc(rowptr: np.ndarray, col: np.ndarray, rank: int, world: int) -> Tuple[np.ndarray, np.ndarray]:
    """CSC rows of the nodes owned by `rank` (row v // world); ids inside the rows stay global"""
    n = rowptr.size - 1
    owned = np.arange(rank, n, world)
    deg = (rowptr[owned + 1] - rowptr[owned]).astype(np.int64)
    rp = np.zeros(owned.size + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    idx = np.repeat(rowptr[owned] - rp[:-1], deg) + np.arange(int(rp[-1]))
    return rp, col[idx].astype(np.uint32)


def partition_rows(x: np.ndarray, rank: int, world: int) -> np.ndarray:
    return np.ascontiguousarray(x[rank::world])


def _all_to_all_v(payload: torch.Tensor, send_counts, group=None, recv_counts=None):
    """payload rows are grouped by destination rank (send_counts[r] rows each); returns (received rows,
    recv_counts as a list).  Rows may have trailing dims.  send_counts: tensor or list; recv_counts: the list of
    rows each rank will send us when the caller already knows it (the way back of a request/response pair: no
    count exchange, no host synchronisation)."""
    sc = send_counts.tolist() if isinstance(send_counts, torch.Tensor) else list(send_counts)
    if recv_counts is None:
        sct = (send_counts if isinstance(send_counts, torch.Tensor)
               else torch.tensor(sc, dtype=torch.int64, device=payload.device))
        rct = torch.empty_like(sct)
        dist.all_to_all_single(rct, sct, group=group)
        rc = rct.tolist()
    else:
        rc = list(recv_counts)
    out = payload.new_empty((int(sum(rc)),) + tuple(payload.shape[1:]))
    dist.all_to_all_single(out, payload.contiguous(), output_split_sizes=rc, input_split_sizes=sc, group=group)
    return out, rc


class DistKHopSampler:
    """k-hop sampling over the hash-partitioned graph.  `expand(nodes, ksums, f, hash_add) -> (nbr [m, f], cnt [m])`
    runs on the OWNER for requests it receives (nodes/ksums: int64 tensors of uint32 values on `device`)."""

    def __init__(self, expand: Callable, device: torch.device, sampling_seed: int = 42, group=None):
        self.expand = expand
        self.device = device
        self.seed = sampling_seed
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def _hop(self, nodes: torch.Tensor, ksums: torch.Tensor, f: int, hash_add: int):
        """nodes/ksums int64 [m] (INVALID = empty slot) -> nbr int64 [m, f] (INVALID padded), cnt int64 [m]"""
        m = nodes.numel()
        valid = nodes != INVALID
        owner = torch.where(valid, nodes % self.world, torch.zeros_like(nodes))
        # invalid slots are not sent at all
        idx = torch.nonzero(valid, as_tuple=False).view(-1)
        order = idx[torch.argsort(owner[idx], stable=True)]
        send_counts = torch.bincount(owner[idx], minlength=self.world).to(torch.int64)
        req = torch.stack([nodes[order], ksums[order]], dim=1)
        sc = send_counts.tolist()
        got, recv_counts = _all_to_all_v(req, sc, self.group)
        nbr_loc, cnt_loc = self.expand(got[:, 0].contiguous(), got[:, 1].contiguous(), f, hash_add)
        resp = torch.cat([nbr_loc.view(-1, f).to(torch.int64), cnt_loc.view(-1, 1).to(torch.int64)], dim=1)
        back, _ = _all_to_all_v(resp, recv_counts, self.group, recv_counts=sc)  # one answer per request
        nbr = torch.full((m, f), INVALID, dtype=torch.int64, device=nodes.device)
        cnt = torch.zeros(m, dtype=torch.int64, device=nodes.device)
        nbr[order] = back[:, :f]
        cnt[order] = back[:, f]
        return nbr, cnt

    def sample_khop(self, roots: torch.Tensor, fanouts: Sequence[int]):
        """roots int64 [b] -> (nbr[k] int64 [slots_k], cnt[k] int64 [parents_k]) in the tree layout of
        include/gigl_hip.h — identical to what a single process samples on the whole graph"""
        nodes = roots.to(self.device).to(torch.int64)
        ksums = nodes.clone()
        out_nbr: List[torch.Tensor] = []
        out_cnt: List[torch.Tensor] = []
        for k, f in enumerate(fanouts):
            hash_add = (self.seed * (k + 1)) & 0xFFFFFFFF
            nbr, cnt = self._hop(nodes, ksums & 0xFFFFFFFF, int(f), hash_add)
            out_nbr.append(nbr.view(-1))
            out_cnt.append(cnt)
            child_k = (ksums.view(-1, 1) + nbr) & 0xFFFFFFFF  # K accumulates along the path (uint32 wrap)
            nodes = nbr.view(-1)
            ksums = torch.where(nodes != INVALID, child_k.view(-1), torch.zeros_like(nodes))
        return out_nbr, out_cnt


class HipDistKHopSampler:
    """k-hop sampling over the hash-partitioned graph with the whole exchange on the device: per hop the frontier is
    bucketed by owner into fixed-capacity buffers (gigl_frontier_bucket), ONE equal-split all_to_all carries the
    requests, the owners answer with gigl_expand_frontier, one all_to_all carries the answers, and
    gigl_frontier_scatter writes the tree layout — no sort, no split sizes, no host synchronisation anywhere in a
    hop (DistKHopSampler above is the backend-agnostic torch version the CPU tests drive).
    Buckets hold `cap` requests per peer (cap = m for world <= 2, else 1.5 m / world + 512: owner(v) = v % world
    is a uniform hash); `overflow` (device int32) becomes non-zero if a bucket overflowed: check it when the step
    is synchronised anyway and redo the batch with slack=world."""

    def __init__(self, eng, world: int, max_window_end: int = -1, sampling_seed: int = 42, group=None,
                 slack: float = 1.5):
        self.eng, self.world, self.mwe, self.seed, self.group, self.slack = eng, world, max_window_end, sampling_seed, group, slack
        self._bufs = {}
        self.overflow = torch.zeros(1, dtype=torch.int32, device=eng.device)

    def capacity(self, m: int) -> int:
        if self.world <= 2 or self.slack >= self.world:
            return max(m, 1)
        return max(1, min(m, int(self.slack * m / self.world) + 512))

    def _hop_buffers(self, k: int, m: int, f: int):
        key = (k, m, f)
        b = self._bufs.get(key)
        if b is None:
            dev, w, cap = self.eng.device, self.world, self.capacity(m)
            i32 = dict(dtype=torch.int32, device=dev)
            b = dict(cap=cap, req=torch.empty((w, 2, cap), **i32), got=torch.empty((w, 2, cap), **i32),
                     slot_idx=torch.empty((w, cap), **i32), counts=torch.zeros(w + 1, **i32),
                     back=torch.empty((w, cap, f), **i32), child_ksums=torch.empty(m * f, **i32))
            self._bufs[key] = b
        return b

    # the three phases of a hop (split so that a test can play all ranks of a world inside one process)
    def bucket(self, k: int, nodes: torch.Tensor, ksums: Optional[torch.Tensor], f: int) -> torch.Tensor:
        b = self._hop_buffers(k, int(nodes.numel()), f)
        self.eng.frontier_bucket(nodes, ksums, self.world, b["cap"], b["req"], b["slot_idx"], b["counts"])
        torch.maximum(self.overflow, b["counts"][self.world:], out=self.overflow)
        return b["req"]

    def serve(self, got: torch.Tensor, k: int, f: int) -> torch.Tensor:
        """owner side: requests received from every peer [world, 2, cap] -> answers [world, cap, f]"""
        cap = int(got.shape[2])
        nodes = got[:, 0, :].reshape(-1).contiguous()
        ksums = got[:, 1, :].reshape(-1).contiguous()
        hash_add = (self.seed * (k + 1)) & 0xFFFFFFFF
        nbr, _ = self.eng.expand_frontier(nodes, ksums, f, hash_add, self.world, self.mwe)
        return nbr.view(self.world, cap, f)

    def scatter(self, k: int, back: torch.Tensor, parent_ksums: torch.Tensor, m: int, f: int, out_nbr: torch.Tensor,
                out_cnt: torch.Tensor) -> torch.Tensor:
        b = self._hop_buffers(k, m, f)
        self.eng.frontier_scatter(back.contiguous(), b["slot_idx"], b["counts"], parent_ksums, m, self.world, b["cap"],
                                  f, out_nbr, out_cnt, b["child_ksums"])
        return b["child_ksums"]

    def sample_khop(self, roots: torch.Tensor, fanouts: Sequence[int], tree=None):
        """roots: int32 device tensor (uint32 ids).  Fills `tree` (HipEngine.alloc_tree) or fresh tensors; returns
        (nbr list, cnt list) of int32 device tensors in the tree layout of include/gigl_hip.h."""
        dev = self.eng.device
        nodes, ksums, parent_k = roots.contiguous(), None, roots.contiguous()
        m = int(roots.numel())
        out_nbr, out_cnt = [], []
        for k, f in enumerate(int(v) for v in fanouts):
            b = self._hop_buffers(k, m, f)
            req = self.bucket(k, nodes, ksums, f)
            dist.all_to_all_single(b["got"], req, group=self.group)
            resp = self.serve(b["got"], k, f)
            dist.all_to_all_single(b["back"], resp.contiguous(), group=self.group)
            nbr = tree.nbr[k] if tree is not None else torch.empty(m * f, dtype=torch.int32, device=dev)
            cnt = tree.cnt[k] if tree is not None else torch.empty(m, dtype=torch.int32, device=dev)
            child_k = self.scatter(k, b["back"], parent_k, m, f, nbr, cnt)
            out_nbr.append(nbr)
            out_cnt.append(cnt)
            nodes, ksums, parent_k, m = nbr, child_k, child_k, m * f
        return out_nbr, out_cnt


class HipFeaturePuller:
    """feature pull of a batch's UNIQUE node ids from their owners, bucketed on the device (gigl_frontier_bucket):
    ids go out in one equal-split all_to_all of fixed-capacity buckets, the per-peer counts in a second tiny one, and
    ONE host read of those counts (the only synchronisation of a sharded step) gives the split sizes of the row
    exchange — rows are never padded.  Phases are separate methods so that a test can play every rank of a world."""

    def __init__(self, eng, world: int, local_rows: torch.Tensor, cap_ids: int, group=None, slack: float = 1.5):
        self.eng, self.world, self.x_local, self.group = eng, world, local_rows, group
        m = int(cap_ids)
        self.m = m
        self.cap = max(m, 1) if (world <= 2 or slack >= world) else max(1, min(m, int(slack * m / world) + 512))
        i32 = dict(dtype=torch.int32, device=eng.device)
        self.req = torch.empty((world, 2, self.cap), **i32)
        self.got = torch.empty((world, 2, self.cap), **i32)
        self.slot_idx = torch.empty((world, self.cap), **i32)
        self.counts = torch.zeros(world + 1, **i32)
        self.recv_counts = torch.zeros(world + 1, **i32)
        self._arange = torch.arange(m, **i32)

    def request(self, ids: torch.Tensor, n_valid_dev: torch.Tensor) -> torch.Tensor:
        """ids: int32 [cap_ids] device (uint32 payload), the first *n_valid_dev are real -> request buckets"""
        masked = torch.where(self._arange < n_valid_dev, ids[: self.m], torch.full_like(self._arange, -1))
        self.eng.frontier_bucket(masked, None, self.world, self.cap, self.req, self.slot_idx, self.counts)
        return self.req

    def serve(self, got: torch.Tensor, recv_counts: Sequence[int]) -> torch.Tensor:
        """owner side: the feature rows of the ids received from every peer, in request order"""
        ids = torch.cat([got[r, 0, : int(recv_counts[r])] for r in range(self.world)]).to(torch.int64) & 0xFFFFFFFF
        return self.x_local.index_select(0, ids // self.world)

    def place(self, back: torch.Tensor, send_counts: Sequence[int], n_rows: int) -> torch.Tensor:
        """rows received for our requests (bucket order) -> x[local id] for the batch's n_rows union-graph nodes"""
        slots = torch.cat([self.slot_idx[r, : int(send_counts[r])] for r in range(self.world)]).to(torch.int64)
        x = torch.empty((n_rows, back.shape[1]), dtype=back.dtype, device=back.device)
        x.index_copy_(0, slots, back)
        return x

    def place_index(self, send_counts: Sequence[int]) -> torch.Tensor:
        """instead of copying the received rows into node order (place): pos[local id] = row of `back` that holds the
        node's features — handed to the first layer as gather_ids (HipBatch.x_index), so the rows are read once, where
        they arrived"""
        slots = torch.cat([self.slot_idx[r, : int(send_counts[r])] for r in range(self.world)]).to(torch.int64)
        pos = torch.zeros(self.m, dtype=torch.int32, device=slots.device)
        pos[slots] = torch.arange(slots.numel(), dtype=torch.int32, device=slots.device)
        return pos

    def pull(self, ids: torch.Tensor, n_valid_dev: torch.Tensor) -> torch.Tensor:
        req = self.request(ids, n_valid_dev)
        dist.all_to_all_single(self.got, req, group=self.group)
        dist.all_to_all_single(self.recv_counts[: self.world], self.counts[: self.world], group=self.group)
        host = torch.cat([self.counts, self.recv_counts[: self.world], n_valid_dev.view(1).to(torch.int32)]).tolist()
        sc, overflow, rc, n_rows = host[: self.world], host[self.world], host[self.world + 1: 2 * self.world + 1], host[-1]
        if overflow:
            raise RuntimeError("feature-pull bucket overflow: raise slack (owner(v) = v % world is badly skewed here)")
        rows = self.serve(self.got, rc)
        back, _ = _all_to_all_v(rows, rc, self.group, recv_counts=sc)
        return self.place(back, sc, n_rows)


class Comm:
    """a communicator of the library (include/gigl_hip.h `gigl_comm`): the transport of the sharded plan's exchanges.
    Comm.rccl: RCCL over xGMI (production; one per ctx); Comm.local: every rank of a world inside this process on one
    device (tests / single-process drivers); Comm.callback: the caller moves the bytes (e.g. gloo through
    torch.distributed — see torch_exchange)"""

    def __init__(self, eng, handle, keep=None):
        self.eng, self._h, self._keep = eng, handle, keep
        rank, world, kind = C.c_int32(), C.c_int32(), C.c_int32()
        _check(eng._lib.gigl_comm_info(handle, C.byref(rank), C.byref(world), C.byref(kind)), eng._ctx)
        self.rank, self.world, self.kind = rank.value, world.value, kind.value

    @staticmethod
    def unique_id() -> bytes:
        from . import _lib
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _check(_lib.load().gigl_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def rccl(eng, rank: int, world: int, unique_id: bytes) -> "Comm":
        h = C.c_void_p()
        _check(eng._lib.gigl_dist_init(eng._ctx, rank, world, C.c_char_p(unique_id), C.byref(h)), eng._ctx)
        return Comm(eng, h)

    @staticmethod
    def rccl_from_torch(eng, group=None) -> "Comm":
        """RCCL communicator for the ranks of a torch.distributed group: rank 0's id travels through the group"""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [Comm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return Comm.rccl(eng, rank, world, box[0])

    @staticmethod
    def from_torch(eng, group=None) -> "Comm":
        """the communicator for the ranks of a torch.distributed group: RCCL over xGMI when the group's backend is nccl
        (one GPU per rank: the production transport, issued by the library on the plan's stream); any other backend
        (gloo: ranks sharing one GPU in tests, or no device buffers) moves the blocks through the host callback"""
        if dist.get_backend(group) == "nccl":
            return Comm.rccl_from_torch(eng, group)
        return Comm.callback(eng, dist.get_rank(group), dist.get_world_size(group), torch_exchange(eng, group))

    @staticmethod
    def local(engines) -> List["Comm"]:
        world = len(engines)
        ctxs = (C.c_void_p * world)(*[e._ctx for e in engines])
        out = (C.c_void_p * world)()
        _check(engines[0]._lib.gigl_dist_init_local(ctxs, world, out), engines[0]._ctx)
        return [Comm(engines[r], C.c_void_p(out[r])) for r in range(world)]

    @staticmethod
    def callback(eng, rank: int, world: int, fn: Callable[[int, int, int], None]) -> "Comm":
        """fn(send_ptr, recv_ptr, bytes_per_peer) with DEVICE addresses; exceptions fail the exchange"""
        from . import _lib

        def tramp(_user, send, recv, nbytes):
            try:
                fn(send, recv, nbytes)
                return 0
            except Exception:  # noqa: BLE001 — reported through the C status
                import traceback
                traceback.print_exc()
                return 1
        cb = _lib.EXCHANGE_FN(tramp)
        h = C.c_void_p()
        _check(eng._lib.gigl_dist_init_callback(eng._ctx, rank, world, cb, None, C.byref(h)), eng._ctx)
        return Comm(eng, h, keep=cb)

    def all_to_all(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        assert send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous()
        nbytes = send.numel() * send.element_size()
        assert nbytes == recv.numel() * recv.element_size() and nbytes % self.world == 0
        _check(self.eng._lib.gigl_comm_all_to_all(self._h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                                                  nbytes // self.world), self.eng._ctx)

    def flush_local(self) -> None:
        _check(self.eng._lib.gigl_comm_flush_local(self._h), self.eng._ctx)

    def set_fixed_blocks(self, on: bool) -> None:
        """whole row blocks instead of count-sized ones (gigl_comm_set_fixed_blocks): no host read in a step"""
        _check(self.eng._lib.gigl_comm_set_fixed_blocks(self._h, 1 if on else 0), self.eng._ctx)

    def traffic(self) -> Tuple[int, int]:
        """(bytes moved, bytes full blocks would have moved) to OTHER ranks since creation (gigl_comm_traffic): the
        sharded plans send only the requested rows of each feature-row block"""
        moved, full = C.c_int64(0), C.c_int64(0)
        _check(self.eng._lib.gigl_comm_traffic(self._h, C.byref(moved), C.byref(full)), self.eng._ctx)
        return int(moved.value), int(full.value)

    def close(self) -> None:
        if self._h:
            self.eng._lib.gigl_comm_destroy(self._h)
            self._h = None


def torch_exchange(eng, group=None) -> Callable[[int, int, int], None]:
    """Comm.callback transport over a torch.distributed group of ANY backend: blocks are staged through host
    memory (gloo) — a test / fallback transport, the production one is Comm.rccl"""
    world = dist.get_world_size(group)

    def fn(send_ptr, recv_ptr, nbytes):
        from ._lib import LOC_DEVICE, LOC_HOST
        total = nbytes * world
        hs = np.empty(total, dtype=np.uint8)
        _check(eng._lib.gigl_memcpy(eng._ctx, C.c_void_p(hs.ctypes.data), LOC_HOST, C.c_void_p(send_ptr), LOC_DEVICE,
                                    total), eng._ctx)
        ts, tr = torch.from_numpy(hs), torch.empty(total, dtype=torch.uint8)
        dist.all_to_all_single(tr, ts, group=group) if dist.get_backend(group) != "gloo" else _gloo_all_to_all(tr, ts, world, group)
        hr = tr.numpy()
        _check(eng._lib.gigl_memcpy(eng._ctx, C.c_void_p(recv_ptr), LOC_DEVICE, C.c_void_p(hr.ctypes.data), LOC_HOST,
                                    total), eng._ctx)
    return fn


def _gloo_all_to_all(out: torch.Tensor, inp: torch.Tensor, world: int, group=None) -> None:
    """all_to_all over gloo (which has no all_to_all_single for every build): pairwise isend / irecv"""
    rank = dist.get_rank(group)
    n = inp.numel() // world
    reqs = []
    for p in range(world):
        if p == rank:
            out[p * n:(p + 1) * n] = inp[p * n:(p + 1) * n]
            continue
        reqs.append(dist.isend(inp[p * n:(p + 1) * n].contiguous(), dst=p, group=group))
        reqs.append(dist.irecv(out[p * n:(p + 1) * n], src=p, group=group))
    for r in reqs:
        r.wait()


class DistSagePlan:
    """the sharded batch plan (include/gigl_hip.h `gigl_dist_plan_*`): sample over the hash-partitioned graph ->
    union graph -> feature pull -> GraphSAGE forward -> one row per root, every exchange issued by the library.
    `eng` holds THIS rank's shard: load_csc(partition_csc(...)) and load_features(partition_rows(...))."""

    def __init__(self, comm: Comm, weights, biases, b: int, fanouts: Sequence[int], act_last: bool = False,
                 group_roots: Optional[int] = None, project_on_owner: bool = False, pull_cap: int = 0,
                 hop_slack: float = 0.0, max_window_end: int = -1, projected: Optional[torch.Tensor] = None,
                 pull_cap_b: int = 0, aggr: str = "mean", staged: bool = False, peer_direct: bool = False,
                 peer_sample: bool = False):
        """aggr: the SAGE layers' reduction ("mean" | "sum" | "max"; "max" pulls raw rows: not with project_on_owner /
        projected).  projected: this rank's pre-projected rows (HipEngine.project_features of the SHARD's table with weights[0]:
        [shard rows, 2*out] fp32) — the pull moves W_l x rows, the first layer is one reduction (gigl_dist_plan_opts.
        projected); recompute and rebuild the plan after a weight update.
        peer_direct: the peer-mapped route (gigl_dist_plan_opts.peer_direct): rows are read where they live, from the
        owners' tables mapped into this process — hand them over with set_peer_tables / map_peer_tables before the first step.
        peer_sample (with peer_direct): the peer-sampled route (gigl_dist_plan_opts.peer_sample): the ranks' graph shards are
        mapped too (set_peer_graphs) and every rank expands its own frontier over them — a step without any exchange.
        staged: the plan serves TRAINING batches (gigl_dist_plan_opts.staged): sample_and_pull + batch_tensors hand out the
        batch union graph and its dense feature matrix; raw rows, every union node numbered"""
        from . import _lib
        eng = comm.eng
        assert eng._graph is not None and eng._feat is not None, "load this rank's shard first"
        L = len(fanouts)
        self.comm, self.eng, self.b, self.fanouts = comm, eng, int(b), [int(f) for f in fanouts]
        self.dims = [int(weights[0].shape[1]) // 2] + [int(w.shape[0]) for w in weights]
        self._lib = eng._lib
        self._plan = C.c_void_p()
        w_arr, b_arr = self._ptr_arrays(weights, biases)
        o = _lib.GiglDistPlanOpts()
        o.group_roots = int(group_roots or b)
        o.project_on_owner = 1 if project_on_owner else 0
        o.pull_cap, o.hop_slack, o.max_window_end = int(pull_cap), float(hop_slack), int(max_window_end)
        o.staged = 1 if staged else 0
        o.peer_direct = 1 if peer_direct else 0
        o.peer_sample = 1 if peer_sample else 0
        self.peer_direct, self.peer_sample = bool(peer_direct), bool(peer_sample)
        self._peer_keep = None
        self.staged = bool(staged)
        assert not (staged and (projected is not None or project_on_owner)), "staged batches pull raw rows"
        self.projected = projected
        if projected is not None:
            assert projected.is_cuda and projected.dtype == torch.float32 and projected.is_contiguous() and \
                projected.shape[1] == 2 * self.dims[1]
            o.projected = projected.data_ptr()
            o.pull_cap_b = int(pull_cap_b)
        fo = (C.c_int32 * L)(*self.fanouts)
        dims = (C.c_int32 * (L + 1))(*self.dims)
        _check(self._lib.gigl_dist_plan_create(comm._h, eng._graph, eng._feat, self.b, fo, L, dims, w_arr, b_arr,
                                               1 if act_last else 0, C.byref(o), C.byref(self._plan)), eng._ctx)
        if aggr != "mean":
            _check(self._lib.gigl_dist_plan_set_aggr(self._plan, _lib.AGGR[aggr]), eng._ctx)
        n = C.c_int32()
        _check(self._lib.gigl_dist_plan_phases(self._plan, C.byref(n)), eng._ctx)
        self.n_phases = n.value

    def _ptr_arrays(self, weights, biases):
        L = len(weights)
        ws = [w.detach().to(device=self.eng.device, dtype=torch.float32).contiguous() for w in weights]
        bs = [None if x is None else x.detach().to(device=self.eng.device, dtype=torch.float32).contiguous()
              for x in biases]
        self._keep = (ws, bs)  # the plan borrows these device buffers
        return ((C.c_void_p * L)(*[w.data_ptr() for w in ws]),
                (C.c_void_p * L)(*[(x.data_ptr() if x is not None else None) for x in bs]))

    def set_weights(self, weights, biases) -> None:
        w_arr, b_arr = self._ptr_arrays(weights, biases)
        _check(self._lib.gigl_dist_plan_set_weights(self._plan, w_arr, b_arr), self.eng._ctx)

    def new_out(self) -> torch.Tensor:
        return torch.empty((self.b, self.dims[-1]), dtype=torch.float32, device=self.eng.device)

    def set_hot_rows(self, hot_ids: Optional[torch.Tensor], hot_rows: Optional[torch.Tensor]) -> None:
        """replicated hot rows (gigl_dist_plan_set_hot_rows): `hot_ids` distinct GLOBAL node ids, `hot_rows` their
        feature rows [n_hot, d] in the shard's feature dtype — the same set on every rank; such rows are read locally
        and never pulled.  None / empty clears the set."""
        n = 0 if hot_ids is None else int(hot_ids.numel())
        if n == 0:
            _check(self._lib.gigl_dist_plan_set_hot_rows(self._plan, None, 0, None), self.eng._ctx)
            self._hot = None
            return
        ids = hot_ids.to(device=self.eng.device, dtype=torch.int32).contiguous()
        rows = hot_rows.to(device=self.eng.device).contiguous()
        if getattr(self, "projected", None) is not None:  # pre-projected plan: the replicas are W_l x rows
            assert rows.dtype == torch.float32 and rows.shape[1] == self.dims[1], "hot rows of a pre-projected plan: [n, out] fp32"
        else:
            assert rows.shape[1] == self.dims[0]
        assert rows.shape[0] == n
        _check(self._lib.gigl_dist_plan_set_hot_rows(self._plan, C.c_void_p(ids.data_ptr()), n,
                                                     C.c_void_p(rows.data_ptr())), self.eng._ctx)
        self._hot = (ids, rows)  # the plan borrows the rows

    # ---- peer-mapped route
    def own_table(self) -> int:
        """device address of the table of this rank the other ranks read: its pre-projected rows, or its feature rows"""
        return int(self.projected.data_ptr()) if self.projected is not None else int(self.eng._feat_ptr.value)

    def set_peer_tables(self, tables: Sequence) -> None:
        """tables[r]: rank r's table as a device tensor or a raw device address valid in THIS process (the ranks of an
        in-process group: each other's tensors; separate processes: the addresses map_peer_tables opens)"""
        assert self.peer_direct and len(tables) == self.comm.world
        ptrs = [int(t.data_ptr()) if isinstance(t, torch.Tensor) else int(t) for t in tables]
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        _check(self._lib.gigl_dist_plan_set_peer_tables(self._plan, arr), self.eng._ctx)
        self._peer_keep = list(tables)  # (tensors stay alive with the plan)

    def own_graph(self) -> Tuple[int, int]:
        """device addresses (rowptr, col) of this rank's CSC shard (gigl_graph_device_ptrs)"""
        rp, cl = C.c_void_p(), C.c_void_p()
        _check(self._lib.gigl_graph_device_ptrs(self.eng._graph, C.byref(rp), C.byref(cl)), self.eng._ctx)
        return int(rp.value), int(cl.value)

    def set_peer_graphs(self, rowptrs: Sequence[int], cols: Sequence[int]) -> None:
        """rowptrs[r] / cols[r]: rank r's CSC shard as device addresses valid in THIS process (own_graph of the ranks of an
        in-process group; share_tables of each array between processes)"""
        assert self.peer_sample and len(rowptrs) == len(cols) == self.comm.world
        ra = (C.c_void_p * len(rowptrs))(*[int(v) for v in rowptrs])
        ca = (C.c_void_p * len(cols))(*[int(v) for v in cols])
        _check(self._lib.gigl_dist_plan_set_peer_graphs(self._plan, ra, ca), self.eng._ctx)

    @staticmethod
    def share_tables(eng, table, group=None) -> Tuple[list, list]:
        """every rank's `table` mapped into this process: export (hipIpc handle of the allocation + offset), all_gather
        over `group` (any backend: the handles are host bytes), open the peers'.  `table`: a device tensor or a device address
        (DistSagePlan.own_table).  Returns (addresses by rank, the opened bases for close_shared).  One call per table and process; the addresses
        serve every plan of the process."""
        from . import _lib
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        h = (C.c_uint8 * _lib.IPC_HANDLE_BYTES)()
        off = C.c_int64()
        addr = int(table.data_ptr()) if isinstance(table, torch.Tensor) else int(table)
        _check(eng._lib.gigl_ipc_export(eng._ctx, C.c_void_p(addr), h, C.byref(off)), eng._ctx)
        mine = bytes(h) + int(off.value).to_bytes(8, "little")
        got = [None] * world
        dist.all_gather_object(got, mine, group=group)
        addrs, bases = [], []
        for r in range(world):
            if r == rank:
                addrs.append(addr)
                continue
            hb = (C.c_uint8 * _lib.IPC_HANDLE_BYTES).from_buffer_copy(got[r][:_lib.IPC_HANDLE_BYTES])
            o = int.from_bytes(got[r][_lib.IPC_HANDLE_BYTES:], "little")
            base, ptr = C.c_void_p(), C.c_void_p()
            _check(eng._lib.gigl_ipc_open(eng._ctx, hb, o, C.byref(base), C.byref(ptr)), eng._ctx)
            addrs.append(int(ptr.value))
            bases.append(int(base.value))
        return addrs, bases

    @staticmethod
    def close_shared(eng, bases: Sequence[int]) -> None:
        for b in bases:
            _check(eng._lib.gigl_ipc_close(eng._ctx, C.c_void_p(b)), eng._ctx)

    def run(self, roots: torch.Tensor, out: Optional[torch.Tensor] = None, sampling_seed: int = 42) -> torch.Tensor:
        assert roots.is_cuda and roots.dtype == torch.int32 and roots.numel() == self.b and roots.is_contiguous()
        out = out if out is not None else self.new_out()
        _check(self._lib.gigl_dist_plan_run(self._plan, C.c_void_p(roots.data_ptr()), sampling_seed,
                                            C.c_void_p(out.data_ptr())), self.eng._ctx)
        return out

    # ---- staged plans: training batches of a hash-partitioned graph
    def sample_and_pull(self, roots: torch.Tensor, sampling_seed: int = 42) -> None:
        """the phases before the forward — per-hop requests / answers, union graph, feature pull — of one step on an
        RCCL / callback communicator (every rank calls it once per step)"""
        assert self.staged and roots.is_cuda and roots.dtype == torch.int32 and roots.numel() == self.b and roots.is_contiguous()
        scratch = self.__dict__.setdefault("_scratch_out", self.new_out())
        for ph in range(self.n_phases - 1):
            _check(self._lib.gigl_dist_plan_phase(self._plan, ph, C.c_void_p(roots.data_ptr()), sampling_seed,
                                                  C.c_void_p(scratch.data_ptr())), self.eng._ctx)

    @staticmethod
    def sample_and_pull_local(plans: Sequence["DistSagePlan"], roots: Sequence[torch.Tensor], sampling_seed: int = 42) -> None:
        """sample_and_pull of every rank of an in-process group (Comm.local), phase by phase"""
        for ph in range(plans[0].n_phases - 1):
            for p, r in zip(plans, roots):
                scratch = p.__dict__.setdefault("_scratch_out", p.new_out())
                _check(p._lib.gigl_dist_plan_phase(p._plan, ph, C.c_void_p(r.data_ptr()), sampling_seed,
                                                   C.c_void_p(scratch.data_ptr())), p.eng._ctx)
            plans[0].comm.flush_local()

    def batch_tensors(self) -> dict:
        """the batch of the step run last (sample_and_pull) as device tensors of the caller's: x [cap_nodes, d] fp32 (rows
        >= meta[0] unset), rowptr / rowend [cap_nodes + 1], col [cap_edges], root_local [b], meta [16], nodes [cap_nodes]
        (global ids) — copies on the plan's stream: the plan's buffers are reused by the next step"""
        from ._lib import GIGL_META_LEN, GiglTree, GiglUnion
        t, u = GiglTree(), GiglUnion()
        _check(self._lib.gigl_dist_plan_buffers(self._plan, C.byref(t), C.byref(u)), self.eng._ctx)
        dev, cn, ce = self.eng.device, int(u.cap_nodes), int(u.cap_edges)
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
        out = dict(x=torch.empty((cn, self.dims[0]), dtype=torch.float32, device=dev), rowptr=i32(cn + 1), rowend=i32(cn + 1),
                   col=i32(max(ce, 1)), root_local=i32(self.b), meta=i32(GIGL_META_LEN), nodes=i32(cn))
        p = lambda k: C.c_void_p(out[k].data_ptr())
        _check(self._lib.gigl_dist_plan_batch_features(self._plan, p("x")), self.eng._ctx)
        _check(self._lib.gigl_dist_plan_batch_graph(self._plan, p("rowptr"), p("rowend"), p("col"), p("root_local"), p("meta"),
                                                    p("nodes")), self.eng._ctx)
        return out

    @staticmethod
    def run_local(plans: Sequence["DistSagePlan"], roots: Sequence[torch.Tensor],
                  outs: Optional[Sequence[torch.Tensor]] = None, sampling_seed: int = 42):
        """one step of every rank of an in-process group (Comm.local), phase by phase"""
        world = len(plans)
        outs = list(outs) if outs is not None else [p.new_out() for p in plans]
        pa = (C.c_void_p * world)(*[p._plan for p in plans])
        ra = (C.c_void_p * world)(*[r.data_ptr() for r in roots])
        oa = (C.c_void_p * world)(*[o.data_ptr() for o in outs])
        _check(plans[0]._lib.gigl_dist_plan_run_local(pa, world, ra, sampling_seed, oa), plans[0].eng._ctx)
        return outs

    def stats(self, acc: torch.Tensor) -> None:
        assert acc.is_cuda and acc.dtype == torch.int64 and acc.numel() >= 16
        _check(self._lib.gigl_dist_plan_stats(self._plan, C.c_void_p(acc.data_ptr())), self.eng._ctx)

    def bucket_fill(self, acc4: torch.Tensor) -> None:
        """fold the feature-pull bucket fill of the step run last into acc4 (int64 [4], device): max / sum over peers of
        the first pull, max / sum of a pre-projected plan's second pull (gigl_dist_plan_bucket_fill)"""
        assert acc4.is_cuda and acc4.dtype == torch.int64 and acc4.numel() >= 4
        _check(self._lib.gigl_dist_plan_bucket_fill(self._plan, C.c_void_p(acc4.data_ptr())), self.eng._ctx)

    def overflowed(self) -> bool:
        """True when the step run last failed (a hop / row bucket or the workspace overflowed: its output rows are NaN).
        Reads meta[GIGL_META_OVERFLOW] from the device: synchronises the plan's stream — call it where the step's rows
        are consumed anyway"""
        from ._lib import GIGL_META_LEN, GiglTree, GiglUnion, LOC_DEVICE, LOC_HOST
        t, u = GiglTree(), GiglUnion()
        _check(self._lib.gigl_dist_plan_buffers(self._plan, C.byref(t), C.byref(u)), self.eng._ctx)
        meta = np.empty(GIGL_META_LEN, dtype=np.int32)
        _check(self._lib.gigl_memcpy(self.eng._ctx, C.c_void_p(meta.ctypes.data), LOC_HOST, C.c_void_p(u.meta), LOC_DEVICE,
                                     meta.nbytes), self.eng._ctx)
        return bool(meta[8])

    def raise_on_overflow(self) -> None:
        if self.overflowed():
            raise RuntimeError("sharded step failed: a hop / feature-row bucket or the activation workspace overflowed "
                               "(meta[GIGL_META_OVERFLOW]); its rows are NaN — raise hop_slack / pull_cap and redo the batch")

    def buffers_to_host(self):
        """host copies of the last step's tree and union graph (like SagePlan.last_batch_to_host)"""
        from ._lib import GIGL_META_LEN, GiglTree, GiglUnion, LOC_DEVICE, LOC_HOST
        t, u = GiglTree(), GiglUnion()
        _check(self._lib.gigl_dist_plan_buffers(self._plan, C.byref(t), C.byref(u)), self.eng._ctx)

        def d2h(ptr, n, dtype):
            a = np.empty(n, dtype=dtype)
            if n:
                _check(self._lib.gigl_memcpy(self.eng._ctx, C.c_void_p(a.ctypes.data), LOC_HOST, C.c_void_p(ptr),
                                             LOC_DEVICE, a.nbytes), self.eng._ctx)
            return a
        nbr, cnt, parents = [], [], self.b
        for k, f in enumerate(self.fanouts):
            cnt.append(d2h(t.cnt[k], parents, np.int32))
            parents *= f
            nbr.append(d2h(t.nbr[k], parents, np.uint32))
        meta = d2h(u.meta, GIGL_META_LEN, np.int32)
        nn = int(meta[0])
        return dict(nbr=nbr, cnt=cnt, meta=meta, nodes=d2h(u.nodes, nn, np.uint32),
                    rowptr=d2h(u.rowptr, nn + 1, np.int32), rowend=d2h(u.rowend, nn + 1, np.int32),
                    col=d2h(u.col, int(u.cap_edges), np.int32), root_local=d2h(u.root_local, self.b, np.int32))

    def close(self) -> None:
        if getattr(self, "_plan", None):
            self._lib.gigl_dist_plan_destroy(self._plan)
            self._plan = None


class DistGatPlan(DistSagePlan):
    """the sharded plan with GAT layers (gigl_dist_gat_plan_create): sampling over the hash-partitioned graph, union
    graph, pull of the raw feature rows, then the GAT one-call plan's layer stages over the pulled rows.  run /
    run_local / stats / buffers_to_host / overflowed / close are DistSagePlan's."""

    def __init__(self, comm: Comm, weights, att_src, att_dst, biases, heads, channels, b: int, fanouts: Sequence[int],
                 negative_slope: float = 0.2, act_last: bool = False, group_roots: Optional[int] = None,
                 pull_cap: int = 0, hop_slack: float = 0.0, max_window_end: int = -1):
        from . import _lib
        eng = comm.eng
        assert eng._graph is not None and eng._feat is not None, "load this rank's shard first"
        L = len(fanouts)
        assert len(weights) == len(att_src) == len(att_dst) == len(heads) == len(channels) == L
        self.comm, self.eng, self.b, self.fanouts = comm, eng, int(b), [int(f) for f in fanouts]
        self.dims = [int(weights[0].shape[1])] + [int(h) * int(c) for h, c in zip(heads, channels)]
        self._lib = eng._lib
        self._plan = C.c_void_p()
        arrs = self._gat_arrays(weights, att_src, att_dst, biases)
        o = _lib.GiglDistPlanOpts()
        o.group_roots = int(group_roots or b)
        o.project_on_owner = 0
        o.pull_cap, o.hop_slack, o.max_window_end = int(pull_cap), float(hop_slack), int(max_window_end)
        fo = (C.c_int32 * L)(*self.fanouts)
        hd, ch = (C.c_int32 * L)(*[int(h) for h in heads]), (C.c_int32 * L)(*[int(c) for c in channels])
        _check(self._lib.gigl_dist_gat_plan_create(comm._h, eng._graph, eng._feat, self.b, fo, L, hd, ch, *arrs,
                                                   float(negative_slope), 1 if act_last else 0, C.byref(o),
                                                   C.byref(self._plan)), eng._ctx)
        n = C.c_int32()
        _check(self._lib.gigl_dist_plan_phases(self._plan, C.byref(n)), eng._ctx)
        self.n_phases = n.value

    def _gat_arrays(self, weights, att_src, att_dst, biases):
        L = len(weights)
        dev = lambda t: t.detach().to(device=self.eng.device, dtype=torch.float32).contiguous()
        ws, a_s, a_d = [dev(w) for w in weights], [dev(a.reshape(-1)) for a in att_src], [dev(a.reshape(-1)) for a in att_dst]
        bs = [None if x is None else dev(x) for x in biases]
        self._keep = (ws, a_s, a_d, bs)  # the plan borrows these device buffers
        arr = lambda ts: (C.c_void_p * L)(*[(t.data_ptr() if t is not None else None) for t in ts])
        return arr(ws), arr(a_s), arr(a_d), arr(bs)

    def set_weights(self, weights, att_src, att_dst, biases) -> None:
        _check(self._lib.gigl_dist_gat_plan_set_weights(self._plan, *self._gat_arrays(weights, att_src, att_dst, biases)),
               self.eng._ctx)

    def set_hot_rows(self, hot_ids, hot_rows) -> None:
        raise NotImplementedError("replicated hot rows belong to the dense pull bookkeeping of the SAGE plan")


def hip_expand(eng, world: int, max_window_end: int = -1) -> Callable:
    """owner-side expansion on the GPU: adapter from DistKHopSampler's int64 tensors to
    HipEngine.expand_frontier (gigl_expand_frontier on this rank's shard)"""
    def fn(nodes: torch.Tensor, ksums: torch.Tensor, f: int, hash_add: int):
        nbr, cnt = eng.expand_frontier(nodes.to(torch.int32).contiguous(), ksums.to(torch.int32).contiguous(), f,
                                       hash_add, world, max_window_end)
        return nbr.to(torch.int64) & 0xFFFFFFFF, cnt.to(torch.int64)
    return fn


def pull_features(ids: torch.Tensor, local_rows: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """rows of the hash-partitioned feature table for global `ids` (int64 [m], any owners):
    ids -> owners, owners index their shard (row id // world), rows -> requesters, original order restored"""
    owner = ids % world
    order = torch.argsort(owner, stable=True)
    send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
    sc = send_counts.tolist()
    got, recv_counts = _all_to_all_v(ids[order].view(-1, 1), sc, group)
    rows = local_rows[(got.view(-1) // world)]
    back, _ = _all_to_all_v(rows, recv_counts, group, recv_counts=sc)  # one row per requested id
    out = torch.empty_like(back)
    out[order] = back
    return out


def shard_batches(total_batches: int, rank: int, world: int) -> range:
    """root batches are independent units: rank r takes batches r, r+world, ..."""
    return range(rank, total_batches, world)
