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

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

INVALID = 0xFFFFFFFF


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


def _all_to_all_v(payload: torch.Tensor, send_counts: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """payload rows are grouped by destination rank (send_counts[r] rows each); returns (received rows,
    recv_counts).  Rows may have trailing dims."""
    world = dist.get_world_size(group)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    out = payload.new_empty((int(sum(rc)),) + tuple(payload.shape[1:]))
    dist.all_to_all_single(out, payload.contiguous(), output_split_sizes=rc, input_split_sizes=sc, group=group)
    return out, recv_counts


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
        got, recv_counts = _all_to_all_v(req, send_counts, self.group)
        nbr_loc, cnt_loc = self.expand(got[:, 0].contiguous(), got[:, 1].contiguous(), f, hash_add)
        resp = torch.cat([nbr_loc.view(-1, f).to(torch.int64), cnt_loc.view(-1, 1).to(torch.int64)], dim=1)
        back, _ = _all_to_all_v(resp, recv_counts, self.group)
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
    got, recv_counts = _all_to_all_v(ids[order].view(-1, 1), send_counts, group)
    rows = local_rows[(got.view(-1) // world)]
    back, _ = _all_to_all_v(rows, recv_counts, group)
    out = torch.empty_like(back)
    out[order] = back
    return out


def shard_batches(total_batches: int, rank: int, world: int) -> range:
    """root batches are independent units: rank r takes batches r, r+world, ..."""
    return range(rank, total_batches, world)
