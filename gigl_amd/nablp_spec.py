"""Node-anchor-based link prediction task spec (trainer + inferencer plugin) on the HIP path.

Mirror of (paths relative to the reference root):
  NodeAnchorBasedLinkPredictionModelingTaskSpec
        python/gigl/src/common/modeling_task_specs/node_anchor_based_link_prediction_modeling_task_spec.py:66-660
        (kwargs :69-209, init_model :229-316, setup_for_training :319-332, train :334-451, validate :454-571,
         eval :573-624, infer_batch :626-655)
  infer_task_inputs      python/gigl/src/common/modeling_task_specs/utils/infer.py:103-456
  Retrieval task         python/gigl/src/common/models/layers/task.py:108-214
  NodeAnchorBasedLinkPredictionTasks.calculate_losses   task.py:699-758
  EarlyStopper           python/gigl/src/common/modeling_task_specs/utils/early_stop.py:12-59
  KS_FOR_EVAL            python/gigl/src/training/v1/lib/eval_metrics.py
What runs where: the encoder (GraphSAGE over the coalesced batch graph: gather-mean + fp32 MFMA GEMM, with
autograd through gigl_gather_mean_backward) and the inner-product decoder (gigl_linear) are HIP kernels, and so is the
retrieval loss over the [queries x candidates] score matrix (gigl_retrieval_loss, with the count-min-sketch candidate
sampling correction when enabled).
Scope: homogeneous graphs (one condensed node type / edge type) and heterogeneous ones (typed batches through the
native typed collate, HGT / SimpleHGN encoders, first supervision edge type: spec :245-271, infer.py on HeteroData); the
Retrieval, Margin and Softmax tasks — the self-supervised ones (GRACE, BGRL, ...) are not built.
Data: main samples are the NodeAnchorBasedLinkPredictionSample TFRecords and random negatives the
RootedNodeNeighborhood TFRecords the sampler wrote, re-filed into train/val/test by the split generator
(gigl_amd/split_generator.py; datasetMetadata.nodeAnchorBasedLinkPredictionDataset URIs).  Without split outputs
the sampler's files are used directly: root id % 10 (0-7 train, 8 val, 9 test), tiny fixtures (< 100 samples)
whole in every split.
"""
from __future__ import annotations

from copy import deepcopy
from dataclasses import dataclass, field
from itertools import cycle
from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import wire
from .base import (BaseInferencer, BaseTrainer, EvalMetric, EvalMetricsCollection, EvalMetricType, InferBatchResults,
                   hit_rate_at_k, import_obj, mean_reciprocal_rank, no_grad_eval)
from .batches import NodeAnchorBasedLinkPredictionBatch, RootedNodeNeighborhoodBatch, iterate_tfrecord_batches
from .config import GbmlConfigPbWrapper, tfrecord_files
from .link_prediction import LinkPredictionDecoder, LinkPredictionGNN, RetrievalLoss
from .models import GraphSAGE
from .task_specs import _rank_world

KS_FOR_EVAL = [1, 5, 10, 50, 100, 500]


def _strtobool(v) -> bool:
    if isinstance(v, bool):
        return v
    s = str(v).strip().lower()
    if s in ("y", "yes", "t", "true", "on", "1"):
        return True
    if s in ("n", "no", "f", "false", "off", "0"):
        return False
    raise ValueError(f"invalid truth value {v!r}")


# ---- task inputs (python/gigl/src/common/types/task_inputs.py) -------------------------------------
@dataclass
class BatchScores:
    pos_scores: torch.Tensor
    hard_neg_scores: torch.Tensor
    random_neg_scores: torch.Tensor


@dataclass
class BatchCombinedScores:
    repeated_candidate_scores: torch.Tensor
    positive_ids: torch.Tensor
    hard_neg_ids: torch.Tensor
    random_neg_ids: torch.Tensor
    repeated_query_ids: Optional[torch.Tensor]
    num_unique_query_ids: Optional[int]


@dataclass
class BatchEmbeddings:
    query_embeddings: torch.Tensor
    repeated_query_embeddings: Dict[int, torch.Tensor]
    pos_embeddings: Dict[int, torch.Tensor]
    hard_neg_embeddings: Dict[int, torch.Tensor]
    random_neg_embeddings: Dict[int, torch.Tensor]


@dataclass
class NodeAnchorBasedLinkPredictionTaskInputs:
    main_batch: NodeAnchorBasedLinkPredictionBatch
    random_neg_batch: RootedNodeNeighborhoodBatch
    batch_embeddings: Optional[BatchEmbeddings]
    batch_scores: List[Dict[int, BatchScores]] = field(default_factory=list)
    batch_combined_scores: Dict[int, BatchCombinedScores] = field(default_factory=dict)


def infer_task_inputs(model: nn.Module, gbml_config_pb_wrapper: GbmlConfigPbWrapper,
                      main_batch: NodeAnchorBasedLinkPredictionBatch, random_neg_batch: RootedNodeNeighborhoodBatch,
                      should_eval: bool, device: torch.device, need_batch_scores: bool = False
                      ) -> NodeAnchorBasedLinkPredictionTaskInputs:
    """infer.py:103-456 for one condensed edge type: encode both batch graphs, then per root gather the
    positive / hard-negative rows, and build the [sum(num_pos) x (pos | hard_neg | random_neg)] score matrix"""
    from .batches import HeteroNodeAnchorBasedLinkPredictionBatch
    if isinstance(main_batch, HeteroNodeAnchorBasedLinkPredictionBatch):
        return _infer_task_inputs_hetero(model, gbml_config_pb_wrapper, main_batch, random_neg_batch, should_eval, device,
                                         need_batch_scores)
    from .hbm import HbmNablpBatch
    if isinstance(main_batch, HbmNablpBatch):
        return _infer_task_inputs_hbm(model, main_batch, random_neg_batch, should_eval, device, need_batch_scores)
    inner = model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model
    decoder = inner.decode
    cet = 0
    main_emb = model(main_batch.graph.to(device))
    rn_emb = model(random_neg_batch.graph.to(device))
    root_idx = main_batch.root_node_indices.to(device)
    query = main_emb[root_idx]
    rn_root_idx = random_neg_batch.condensed_node_type_to_root_node_indices_map[0].to(device)
    empty = torch.zeros((0,), dtype=torch.float32, device=device)
    rn_root_emb = rn_emb[rn_root_idx] if rn_root_idx.numel() else empty
    want_scores = should_eval or need_batch_scores  # (infer.py:225,279,309: ModelResultType.batch_scores or should_eval)
    rn_scores = decoder(query, rn_root_emb) if (want_scores and rn_root_emb.numel()) else empty

    pos_map = main_batch.pos_supervision_edge_data[cet].root_node_to_target_node_id
    neg_map = main_batch.hard_neg_supervision_edge_data[cet].root_node_to_target_node_id
    pos_ids, neg_ids, rep, nrep = [], [], [], []
    batch_scores: List[Dict[int, BatchScores]] = []
    for r in main_batch.root_node_indices.tolist():
        p, h = pos_map[r], neg_map[r]
        rep.append(p.numel())
        nrep.append(h.numel())
        if p.numel():
            pos_ids.append(p)
        if h.numel():
            neg_ids.append(h)
    d = query.shape[1]
    # (one index upload + one row gather per list, not one per root)
    pos_emb = main_emb[torch.cat(pos_ids).to(device)] if pos_ids else torch.zeros((0, d), device=device)
    neg_emb = main_emb[torch.cat(neg_ids).to(device)] if neg_ids else torch.zeros((0, d), device=device)
    if want_scores:
        batch_scores = _per_root_scores(decoder, query, pos_emb, neg_emb, rep, nrep, rn_scores, cet, empty)
    rep_t = torch.tensor(rep, device=device)
    rep_query = query.repeat_interleave(rep_t, dim=0)
    cand = torch.cat((pos_emb, neg_emb, rn_root_emb.reshape(-1, d)))
    g_main = main_batch.condensed_node_type_to_subgraph_id_to_global_node_id[0]
    g_rn = random_neg_batch.condensed_node_type_to_subgraph_id_to_global_node_id[0]
    rep_sub_q = main_batch.root_node_indices.repeat_interleave(torch.tensor(rep))

    def to_global(ids, mapping):
        return torch.tensor([mapping[int(v)] for v in ids], dtype=torch.int64, device=device)

    combined = BatchCombinedScores(
        repeated_candidate_scores=decoder(rep_query, cand) if rep_query.numel() else empty,
        positive_ids=to_global(torch.cat(pos_ids) if pos_ids else [], g_main),
        hard_neg_ids=to_global(torch.cat(neg_ids) if neg_ids else [], g_main),
        random_neg_ids=to_global(random_neg_batch.condensed_node_type_to_root_node_indices_map[0], g_rn),
        repeated_query_ids=to_global(rep_sub_q, g_main),
        num_unique_query_ids=int(root_idx.shape[0]))
    return NodeAnchorBasedLinkPredictionTaskInputs(
        main_batch=main_batch, random_neg_batch=random_neg_batch,
        batch_embeddings=BatchEmbeddings(query_embeddings=query, repeated_query_embeddings={cet: rep_query},
                                         pos_embeddings={cet: pos_emb}, hard_neg_embeddings={cet: neg_emb},
                                         random_neg_embeddings={0: rn_root_emb}),
        batch_scores=batch_scores, batch_combined_scores={cet: combined})


def _infer_task_inputs_hbm(model: nn.Module, main_batch, random_neg_batch, should_eval: bool, device: torch.device,
                           need_batch_scores: bool = False) -> NodeAnchorBasedLinkPredictionTaskInputs:
    """infer_task_inputs over batches sampled in HBM (hbm.HbmNablpBatch + the random-negative HbmTrainBatch): the same
    embeddings, scores and ids as over the collated TFRecord batches of the same anchors — the batch graphs hold the
    same nodes and edges — with every index already on the device: no per-root host loop, no id dictionaries.  Hard
    negatives come from user-defined label edges only, which take the TFRecord route."""
    inner = model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model
    decoder = inner.decode
    cet = 0
    hb = main_batch.graph
    ri = main_batch.root_index if main_batch.root_index is not None else hb.root_local.long()
    roots_emb = model(hb)[ri]                                        # [B * (1 + P), d], anchor-major
    d = int(roots_emb.shape[1])
    query = roots_emb.view(main_batch.n_anchors, main_batch.trees_per_anchor, d)[:, 0]
    pos_emb = roots_emb.index_select(0, main_batch.pos_rows)
    rn_emb = model(random_neg_batch.graph)[random_neg_batch.root_node_indices]
    empty = torch.zeros((0,), dtype=torch.float32, device=device)
    neg_emb = torch.zeros((0, d), device=device)
    want_scores = should_eval or need_batch_scores
    rn_scores = decoder(query, rn_emb) if (want_scores and rn_emb.numel()) else empty
    n_pos = main_batch.n_pos.tolist()
    batch_scores: List[Dict[int, BatchScores]] = []
    if want_scores:
        batch_scores = _per_root_scores(decoder, query, pos_emb, neg_emb, n_pos, [0] * len(n_pos), rn_scores, cet, empty)
    rep_t = torch.from_numpy(main_batch.n_pos).to(device)
    rep_query = query.repeat_interleave(rep_t, dim=0, output_size=int(main_batch.pos_rows.numel()))
    cand = torch.cat((pos_emb, rn_emb.reshape(-1, d)))
    anchor_ids = main_batch.root_ids.view(main_batch.n_anchors, main_batch.trees_per_anchor)[:, 0]
    combined = BatchCombinedScores(
        repeated_candidate_scores=decoder(rep_query, cand) if rep_query.numel() else empty,
        positive_ids=main_batch.root_ids.index_select(0, main_batch.pos_rows),
        hard_neg_ids=torch.zeros((0,), dtype=torch.int64, device=device),
        random_neg_ids=torch.from_numpy(np.asarray(random_neg_batch.root_ids, dtype=np.int64)).to(device),
        repeated_query_ids=anchor_ids.repeat_interleave(rep_t, output_size=int(main_batch.pos_rows.numel())),
        num_unique_query_ids=int(main_batch.n_anchors))
    return NodeAnchorBasedLinkPredictionTaskInputs(
        main_batch=main_batch, random_neg_batch=random_neg_batch,
        batch_embeddings=BatchEmbeddings(query_embeddings=query, repeated_query_embeddings={cet: rep_query},
                                         pos_embeddings={cet: pos_emb}, hard_neg_embeddings={cet: neg_emb},
                                         random_neg_embeddings={0: rn_emb}),
        batch_scores=batch_scores, batch_combined_scores={cet: combined})


def _per_root_scores(decoder, query: torch.Tensor, pos_emb: torch.Tensor, neg_emb: torch.Tensor, n_pos: List[int],
                     n_neg: List[int], rn_scores: torch.Tensor, cet: int, empty: torch.Tensor) -> List[Dict[int, BatchScores]]:
    """the evaluation's per-root BatchScores (infer.py:279-310) from TWO projections of the whole batch — every root
    against all positives / all hard negatives — sliced per root, instead of two small products per root (a row of a
    product is the same bits whatever else is in the batch)"""
    b = int(query.shape[0])
    pos_all = decoder(query, pos_emb) if pos_emb.shape[0] else None
    neg_all = decoder(query, neg_emb) if neg_emb.shape[0] else None
    out: List[Dict[int, BatchScores]] = []
    po = no = 0
    for i in range(b):
        p, h = n_pos[i], n_neg[i]
        out.append({cet: BatchScores(
            pos_scores=pos_all[i:i + 1, po:po + p] if p else empty,
            hard_neg_scores=neg_all[i:i + 1, no:no + h] if h else empty,
            random_neg_scores=rn_scores[[i], :] if rn_scores.numel() else empty)})
        po += p
        no += h
    return out


def _infer_task_inputs_hetero(model: nn.Module, cfg: GbmlConfigPbWrapper, main_batch, random_neg_batch,
                              should_eval: bool, device: torch.device, need_batch_scores: bool = False
                              ) -> NodeAnchorBasedLinkPredictionTaskInputs:
    """infer.py:103-456 on typed batches, first supervision edge type (src -> dst): the encoder returns one embedding
    matrix per node type; queries are the roots' rows of the src type, positives / hard negatives rows of the dst type
    (local ids from the typed collate), random negatives the roots of the dst type's RootedNodeNeighborhood batch"""
    inner = model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model
    decoder = inner.decode
    src_t, rel, dst_t = cfg.supervision_edge_types[0]
    name_to_cnt = {v: k for k, v in cfg.condensed_node_type_map.items()}
    cet = [c for c, triple in cfg.condensed_edge_type_map.items() if tuple(triple) == (src_t, rel, dst_t)][0]
    main = model(main_batch.graph.to(device), [src_t, dst_t])
    rn = model(random_neg_batch.graph.to(device), [dst_t])[dst_t]
    root_idx = main_batch.root_node_indices.to(device)
    query = main[src_t][root_idx]
    dst_emb = main[dst_t]
    d = query.shape[1]
    empty = torch.zeros((0,), dtype=torch.float32, device=device)
    rn_root_idx = random_neg_batch.condensed_node_type_to_root_node_indices_map.get(name_to_cnt[dst_t])
    rn_root_emb = rn[rn_root_idx.to(device)] if rn_root_idx is not None and rn_root_idx.numel() else empty
    want_scores = should_eval or need_batch_scores  # (infer.py:225,279,309: ModelResultType.batch_scores or should_eval)
    rn_scores = decoder(query, rn_root_emb) if (want_scores and rn_root_emb.numel()) else empty
    b = int(root_idx.numel())
    none = [torch.zeros(0, dtype=torch.int64)] * b
    pos_l, neg_l = main_batch.pos_targets.get(cet, none), main_batch.hard_neg_targets.get(cet, none)
    batch_scores: List[Dict[int, BatchScores]] = []
    rep = torch.tensor([p.numel() for p in pos_l], dtype=torch.int64)
    pos_ids = torch.cat(pos_l) if b else torch.zeros(0, dtype=torch.int64)
    neg_ids = torch.cat(neg_l) if b else torch.zeros(0, dtype=torch.int64)
    pos_emb = dst_emb[pos_ids.to(device)] if pos_ids.numel() else torch.zeros((0, d), device=device)
    neg_emb = dst_emb[neg_ids.to(device)] if neg_ids.numel() else torch.zeros((0, d), device=device)
    if want_scores:
        batch_scores = _per_root_scores(decoder, query, pos_emb, neg_emb, rep.tolist(), [h.numel() for h in neg_l],
                                        rn_scores, cet, empty)
    rep_query = query.repeat_interleave(rep.to(device), dim=0)
    cand = torch.cat((pos_emb, neg_emb, rn_root_emb.reshape(-1, d)))
    g_main = main_batch.condensed_node_type_to_subgraph_id_to_global_node_id
    g_dst = torch.from_numpy(np.asarray(g_main[name_to_cnt[dst_t]], dtype=np.int64))
    g_src = torch.from_numpy(np.asarray(g_main[name_to_cnt[src_t]], dtype=np.int64))
    rn_globals = torch.tensor([g for t, g in random_neg_batch.root_nodes if t == name_to_cnt[dst_t]], dtype=torch.int64)
    combined = BatchCombinedScores(
        repeated_candidate_scores=decoder(rep_query, cand) if rep_query.numel() else empty,
        positive_ids=g_dst[pos_ids].to(device), hard_neg_ids=g_dst[neg_ids].to(device), random_neg_ids=rn_globals.to(device),
        repeated_query_ids=g_src[main_batch.root_node_indices].repeat_interleave(rep).to(device),
        num_unique_query_ids=b)
    return NodeAnchorBasedLinkPredictionTaskInputs(
        main_batch=main_batch, random_neg_batch=random_neg_batch,
        batch_embeddings=BatchEmbeddings(query_embeddings=query, repeated_query_embeddings={cet: rep_query},
                                         pos_embeddings={cet: pos_emb}, hard_neg_embeddings={cet: neg_emb},
                                         random_neg_embeddings={name_to_cnt[dst_t]: rn_root_emb}),
        batch_scores=batch_scores, batch_combined_scores={cet: combined})


class Retrieval(nn.Module):
    """task.py:108-214; forward -> (summed loss, number of query rows).  With
    should_enable_candidate_sampling_correction the ids of every training batch go through two device-resident count-min
    sketches (positives + hard negatives / random negatives) and logQ of the estimated in-batch probability is taken
    off the logits inside the fused loss"""
    task_name = "Retrieval"
    result_types = ("batch_combined_scores", "batch_embeddings")

    def __init__(self, loss: Optional[nn.Module] = None, temperature: float = 0.07,
                 remove_accidental_hits: bool = True, should_enable_candidate_sampling_correction: bool = False,
                 count_min_sketch_width: int = 10000, count_min_sketch_depth: int = 10):
        super().__init__()
        self.should_enable_candidate_sampling_correction = should_enable_candidate_sampling_correction
        self.loss = RetrievalLoss(loss=loss, temperature=temperature, remove_accidental_hits=remove_accidental_hits)
        if should_enable_candidate_sampling_correction:
            from .count_min_sketch import CountMinSketch
            self.main_batch_cm_sketch = CountMinSketch(width=count_min_sketch_width, depth=count_min_sketch_depth)
            self.random_neg_batch_cm_sketch = CountMinSketch(width=count_min_sketch_width, depth=count_min_sketch_depth)

    def _sampling_probability(self, bcs, device: torch.device) -> torch.Tensor:
        from .count_min_sketch import calculate_in_batch_candidate_sampling_probability as in_batch_q
        main, rn = self.main_batch_cm_sketch, self.random_neg_batch_cm_sketch
        main.add_torch_long_tensor(bcs.positive_ids)
        main.add_torch_long_tensor(bcs.hard_neg_ids)
        rn.add_torch_long_tensor(bcs.random_neg_ids)
        n_main = int(bcs.positive_ids.numel() + bcs.hard_neg_ids.numel())  # the two share a sketch
        return torch.cat((
            in_batch_q(main.estimate_torch_long_tensor(bcs.positive_ids), main.total(), n_main),
            in_batch_q(main.estimate_torch_long_tensor(bcs.hard_neg_ids), main.total(), n_main),
            in_batch_q(rn.estimate_torch_long_tensor(bcs.random_neg_ids), rn.total(), int(bcs.random_neg_ids.numel())),
        )).to(device)

    def forward(self, task_input: NodeAnchorBasedLinkPredictionTaskInputs, gbml_config_pb_wrapper=None,
                should_eval: bool = False, device: torch.device = torch.device("cpu")):
        assert len(task_input.batch_combined_scores) > 0
        assert task_input.batch_embeddings is not None
        running_loss = torch.tensor(0.0, device=device)
        running_batch_size = 0
        for cet, bcs in task_input.batch_combined_scores.items():
            prob = None
            if self.should_enable_candidate_sampling_correction and not should_eval:
                prob = self._sampling_probability(bcs, device)
            rq = task_input.batch_embeddings.repeated_query_embeddings[cet]
            if rq.numel():  # loss.py:333-359
                cand_ids = torch.cat((bcs.positive_ids, bcs.hard_neg_ids, bcs.random_neg_ids)).to(device)
                loss = self.loss.calculate_batch_retrieval_loss(
                    scores=bcs.repeated_candidate_scores, candidate_sampling_probability=prob,
                    query_ids=bcs.repeated_query_ids, candidate_ids=cand_ids, device=device)
                n = int(rq.shape[0])
            else:
                loss, n = torch.tensor(0.0, device=device), 1
            running_loss = running_loss + loss
            running_batch_size += n
        return running_loss, running_batch_size


def _ragged_scores(batch_scores: List[Dict[int, BatchScores]], device: torch.device):
    """the per-root score lists as padded matrices: pos [S, Pmax] + its mask, negatives [S, Nmax] (hard negatives then
    random negatives; padding -inf) for the S (root, edge type) entries that have a positive"""
    rows = [bs for result in batch_scores for bs in result.values() if bs.pos_scores.numel()]
    if not rows:
        return None
    pos = [r.pos_scores.reshape(-1) for r in rows]
    neg = [torch.cat((r.hard_neg_scores.reshape(-1), r.random_neg_scores.reshape(-1))) for r in rows]
    pmax, nmax = max(p.numel() for p in pos), max(max(n_.numel() for n_ in neg), 1)
    P = torch.zeros((len(rows), pmax), device=device)
    M = torch.zeros((len(rows), pmax), dtype=torch.bool, device=device)
    N = torch.full((len(rows), nmax), float("-inf"), device=device)
    for i, (p, n_) in enumerate(zip(pos, neg)):
        P[i, :p.numel()], M[i, :p.numel()] = p, True
        N[i, :n_.numel()] = n_
    return P, M, N


class Margin(nn.Module):
    """task.py:85-105 + MarginLoss (loss.py:21-96): for every root, every (positive, negative) pair contributes
    max(0, margin - pos + neg) — negatives = the root's hard negatives and the batch's random negatives; forward ->
    (summed loss, number of pairs).  Evaluated on padded score matrices, all roots at once."""
    task_name = "Margin"
    result_types = ("batch_scores",)

    def __init__(self, margin: float = 0.5):
        super().__init__()
        self.margin = float(margin)

    def forward(self, task_input: NodeAnchorBasedLinkPredictionTaskInputs, gbml_config_pb_wrapper=None,
                should_eval: bool = False, device: torch.device = torch.device("cpu")):
        assert len(task_input.batch_scores) > 0
        r = _ragged_scores(task_input.batch_scores, device)
        if r is None:
            return torch.tensor(0.0, device=device), 0
        P, M, N = r
        hinge = torch.relu(self.margin - P.unsqueeze(2) + N.unsqueeze(1))  # (-inf padding of N: exactly 0)
        hinge = hinge * M.unsqueeze(2)
        pairs = int((M.sum(1) * torch.isfinite(N).sum(1)).sum().item())
        return hinge.sum(), pairs


class Softmax(nn.Module):
    """task.py:62-82 + SoftmaxLoss (loss.py:99-174): for every (root, positive) the cross-entropy of the positive
    against the root's hard negatives and the batch's random negatives at temperature T; forward -> (summed loss,
    number of positives)."""
    task_name = "Softmax"
    result_types = ("batch_scores",)

    def __init__(self, softmax_temperature: float = 0.07):
        super().__init__()
        self.softmax_temperature = float(softmax_temperature)

    def forward(self, task_input: NodeAnchorBasedLinkPredictionTaskInputs, gbml_config_pb_wrapper=None,
                should_eval: bool = False, device: torch.device = torch.device("cpu")):
        assert len(task_input.batch_scores) > 0
        r = _ragged_scores(task_input.batch_scores, device)
        if r is None:
            return torch.tensor(0.0, device=device), 0
        P, M, N = r
        t = self.softmax_temperature
        neg_lse = torch.logsumexp(N / t, dim=1, keepdim=True)                      # [S, 1] (-inf without negatives)
        lse = torch.logaddexp(P / t, neg_lse)                                      # log(exp(pos) + sum exp(negs))
        loss = ((lse - P / t) * M).sum()
        return loss, int(M.sum().item())


class NodeAnchorBasedLinkPredictionTasks:
    """task.py:699-758: weighted sum of per-sample task losses"""

    def __init__(self) -> None:
        self._task_to_fn_map = nn.ModuleDict()
        self._task_to_weights_map: Dict[str, float] = {}

    def add_task(self, task: nn.Module, weight: float) -> None:
        self._task_to_fn_map[task.task_name] = task
        self._task_to_weights_map[task.task_name] = weight

    @property
    def result_types(self) -> set:
        """what infer_task_inputs has to produce for the registered tasks (task.py result_types)"""
        out = set()
        for task in self._task_to_fn_map.values():
            out |= set(getattr(task, "result_types", ("batch_combined_scores", "batch_embeddings")))
        return out

    def calculate_losses(self, batch_results, gbml_config_pb_wrapper, should_eval: bool, device: torch.device):
        total = torch.tensor(0.0, device=device)
        breakdown: Dict[str, float] = {}
        for name, weight in self._task_to_weights_map.items():
            loss_val, bs = self._task_to_fn_map[name](task_input=batch_results,
                                                      gbml_config_pb_wrapper=gbml_config_pb_wrapper,
                                                      should_eval=should_eval, device=device)
            bs = max(int(bs), 1)  # (a batch without any positive: zero loss, not a division by zero)
            total = total + weight * loss_val / bs
            breakdown[name] = float("{:.3f}".format(float(weight * loss_val.detach()) / bs))
        return total, breakdown


class EarlyStopper:
    """early_stop.py:12-59"""

    def __init__(self, early_stop_criterion: EvalMetricType, early_stop_patience: int):
        supported = [m for m in EvalMetricType.get_all_criteria() if m != "hits"]
        if early_stop_criterion.name not in supported:
            raise NotImplementedError(f"Found invalid early stop criterion {early_stop_criterion.name}. Please make "
                                      f"sure to supply one of {supported}.")
        self.criterion = early_stop_criterion
        self._should_maximize = self.criterion != EvalMetricType.loss
        self.prev_best = float("-inf") if self._should_maximize else float("inf")
        self.early_stop_counter = 0
        self.early_stop_patience = early_stop_patience
        self.best_val_model: Dict[str, Any] = {}

    def has_improved(self, value: float) -> bool:
        return (self._should_maximize and value > self.prev_best) or (
            not self._should_maximize and value < self.prev_best)

    def should_early_stop(self, metrics: Dict[EvalMetricType, Any], model: nn.Module) -> bool:
        value = metrics[self.criterion]
        if self.has_improved(value):
            self.early_stop_counter = 0
            self.prev_best = value
            self.best_val_model = deepcopy(model.state_dict())
        else:
            self.early_stop_counter += 1
        return self.early_stop_counter >= self.early_stop_patience


class HipNodeAnchorLinkPredictionSpec(BaseTrainer, BaseInferencer):
    def __init__(self, **kwargs) -> None:
        gnn_path = str(kwargs.get("gnn_model_class_path", "gigl_amd.models.GraphSAGE"))
        self.gnn_model = import_obj(gnn_path)
        self.hidden_dim = int(kwargs.get("hidden_dim", 16))
        self.num_layers = int(kwargs.get("num_layers", 2))
        self.out_channels = int(kwargs.get("out_channels", 16))
        # encoder-specific arguments (GAT: heads; GAT / EdgeAttrGAT: edge_dim, conv), handed on when the class takes them
        self._encoder_kwargs = {}
        if "num_heads" in kwargs or "heads" in kwargs:
            self._encoder_kwargs["heads"] = int(kwargs.get("num_heads", kwargs.get("heads")))
        if kwargs.get("edge_dim") not in (None, "", "None"):
            self._encoder_kwargs["edge_dim"] = int(kwargs["edge_dim"])
        for k in ("conv", "share_edge_att_message_weight"):
            if k in kwargs:
                self._encoder_kwargs[k] = kwargs[k] if k == "conv" else _strtobool(kwargs[k])
        self.should_l2_normalize_embedding_layer_output = bool(
            kwargs.get("should_l2_normalize_embedding_layer_output", True))
        self.validate_every_n_batches = int(kwargs.get("val_every_num_batches", 20))
        self.num_val_batches = int(kwargs.get("num_val_batches", 10))
        self.num_test_batches = int(kwargs.get("num_test_batches", 100))
        self._optim_cls = import_obj(str(kwargs.get("optim_class_path", "torch.optim.Adam")))
        self._optim_kwargs = {"lr": float(kwargs.get("optim_lr", 5e-3)),
                              "weight_decay": float(kwargs.get("optim_weight_decay", 1e-6))}
        self.clip_grad_norm = float(kwargs.get("clip_grad_norm", 0.0))
        self._lr_scheduler_cls = import_obj(str(kwargs.get("lr_scheduler_name", "torch.optim.lr_scheduler.ConstantLR")))
        self._lr_scheduler_kwargs = {"factor": float(kwargs.get("factor", 1.0)),
                                     "total_iters": int(kwargs.get("total_iters", 10))}
        self.main_sample_batch_size = int(kwargs.get("main_sample_batch_size", 2048))
        self.random_negative_sample_batch_size = int(kwargs.get("random_negative_sample_batch_size", 512))
        self.random_negative_sample_batch_size_for_evaluation = int(
            kwargs.get("random_negative_sample_batch_size_for_evaluation", 512))
        self.early_stopper = EarlyStopper(EvalMetricType[kwargs.get("early_stop_criterion", "loss")],
                                          int(kwargs.get("early_stop_patience", 3)))
        self.tasks = NodeAnchorBasedLinkPredictionTasks()
        task_cls = import_obj(str(kwargs.get("task_path", "gigl_amd.nablp_spec.Retrieval")))
        import inspect
        offered = {"temperature": float(kwargs.get("softmax_temp", 0.07)),
                   "softmax_temperature": float(kwargs.get("softmax_temp", 0.07)),
                   "remove_accidental_hits": _strtobool(kwargs.get("should_remove_accidental_hits", "True")),
                   "margin": float(kwargs.get("margin", 0.5))}
        takes = inspect.signature(task_cls.__init__).parameters
        # (the reference passes temperature / remove_accidental_hits to whatever class task_path names — its Margin and
        # Softmax tasks cannot be built that way; here every task gets the arguments its constructor declares)
        self.tasks.add_task(task_cls(**{k: v for k, v in offered.items() if k in takes}), weight=1.0)
        self._model: Optional[nn.Module] = None
        self._engine = None
        self._cfg: Optional[GbmlConfigPbWrapper] = None
        self.history: List[Dict[str, Any]] = []
        self._kwargs = {k: str(v) for k, v in kwargs.items()}
        self._resident = None       # gigl_amd.hbm.ResidentGraph of the in-HBM route
        self._hbm_anchors = None    # {split: (anchor ids, positives per anchor)}; {} = the TFRecord route was chosen

    @property
    def gbml_config_pb_wrapper(self) -> GbmlConfigPbWrapper:
        if self._cfg is None:
            raise ValueError("gbml_config_pb_wrapper is not initialized before use, run init_model to set.")
        return self._cfg

    @property
    def model(self) -> nn.Module:
        return self._model

    @model.setter
    def model(self, model: nn.Module) -> None:
        self._model = model

    @property
    def supports_distributed_training(self) -> bool:
        return True

    @property
    def supports_hbm_batches(self) -> bool:
        """infer_batch also takes batches sampled in HBM (gigl_amd/hbm.py): encoders with a forward over HipBatches run it,
        the others (GIN, GATv2, Transformer ...) get the same batch as a GraphData built on the device (ResidentGraph.encode)"""
        from .hbm import encoder_takes_hip_batches, encoder_trains_over_graph_data
        inner = self.model.module if hasattr(self.model, "module") else self.model
        if inner is None or (self._cfg is not None and self._cfg.is_heterogeneous):
            return inner is not None and encoder_takes_hip_batches(getattr(inner, "encoder", inner))
        enc = getattr(inner, "encoder", inner)
        return encoder_takes_hip_batches(enc) or encoder_trains_over_graph_data(enc)

    def init_model(self, gbml_config_pb_wrapper: GbmlConfigPbWrapper, state_dict=None) -> nn.Module:
        self._cfg = gbml_config_pb_wrapper
        if gbml_config_pb_wrapper.is_heterogeneous:
            return self._init_hetero_model(gbml_config_pb_wrapper, state_dict)
        in_dim = gbml_config_pb_wrapper.preprocessed_metadata.nodes[0].feature_dim
        import inspect
        accepted = inspect.signature(self.gnn_model.__init__).parameters
        takes_any = any(p.kind == p.VAR_KEYWORD for p in accepted.values())
        extra = {k: v for k, v in self._encoder_kwargs.items() if k in accepted or takes_any}
        encoder = self.gnn_model(
            in_dim=max(in_dim, 1), hid_dim=self.hidden_dim, out_dim=self.out_channels, num_layers=self.num_layers,
            should_l2_normalize_embedding_layer_output=self.should_l2_normalize_embedding_layer_output, **extra)
        model = LinkPredictionGNN(encoder=encoder, decoder=LinkPredictionDecoder())
        if state_dict is not None:
            model.load_state_dict(state_dict)
        self.model = model
        return model

    def _init_hetero_model(self, cfg: GbmlConfigPbWrapper, state_dict=None) -> nn.Module:
        """the heterogeneous branch of init_model (node_anchor_based_link_prediction_modeling_task_spec.py:245-271): the
        encoder class (gnn_model_class_path; default HGT) takes the per-type feature dims, hid / out dims, num_layers,
        num_heads"""
        pm = cfg.preprocessed_metadata
        node_dims = {name: max(pm.nodes[c].feature_dim, 1) for c, name in cfg.condensed_node_type_map.items()}
        edge_dims = {tuple(t): pm.edges[c].feature_dim for c, t in cfg.condensed_edge_type_map.items()}
        gnn = self.gnn_model
        if gnn is GraphSAGE:  # (the homogeneous default was not overridden)
            from .models_hetero import HGT
            gnn = HGT
        import inspect
        accepted = inspect.signature(gnn.__init__).parameters
        dims = dict(hid_dim=self.hidden_dim, out_dim=self.out_channels)
        if "node_hid_dim" in accepted:  # SimpleHGN names its widths per node / edge (heterogeneous.py:122-160)
            dims = dict(node_hid_dim=self.hidden_dim, edge_hid_dim=self.hidden_dim, node_out_dim=self.out_channels,
                        edge_type_dim=int(self._encoder_kwargs.get("edge_type_dim", 16)))
        encoder = gnn(node_type_to_feat_dim_map=node_dims, edge_type_to_feat_dim_map=edge_dims, num_layers=self.num_layers,
                      num_heads=int(self._encoder_kwargs.get("heads", 2)),
                      should_l2_normalize_embedding_layer_output=self.should_l2_normalize_embedding_layer_output, **dims)
        model = LinkPredictionGNN(encoder=encoder, decoder=LinkPredictionDecoder())
        if state_dict is not None:
            model.load_state_dict(state_dict)
        self.model = model
        return model

    def _ensure_engine(self, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("the HIP path needs a GPU device; there is no CPU fallback")
        if self._engine is None:
            from .engine import HipEngine
            self._engine = HipEngine(device.index or 0)
        self._device = device
        inner = self.model.module if hasattr(self.model, "module") else self.model
        inner.encoder.engine = self._engine
        inner.decoder.engine = self._engine
        return self._engine

    def setup_for_training(self):
        self._optimizer = self._optim_cls(params=self.model.parameters(), **self._optim_kwargs)
        self._lr_scheduler = self._lr_scheduler_cls(self._optimizer, **self._lr_scheduler_kwargs)
        self.model.train()

    # ---- data: the in-HBM route (gigl_amd/hbm.py) — main and random-negative batches sampled in HBM
    def _hbm_split(self, cfg: GbmlConfigPbWrapper):
        """-> {split: (anchors, positives per anchor)} when this job's training batches can be sampled in HBM from the
        resident graph, else None (the TFRecord route below).  In HBM: a homogeneous graph whose tables are readable,
        one process, a GraphSAGE encoder with an autograd forward over HipBatches, positives drawn from the graph's own
        edges (user-defined label edges: TFRecord route) and NO split-generator output — a transductive link split
        rewrites every sample's neighbourhood per split (message-passing vs supervision edges), which only the split
        generator's files carry.  The anchors of a split and their order are those of the TFRecord route without split
        files: the sampler's main samples in file order, root id % 10 (0-7 train, 8 val, 9 test), fewer than 100
        samples whole in every split."""
        if self._hbm_anchors is not None:
            return self._hbm_anchors or None
        self._hbm_anchors = {}
        from .hbm import ResidentGraph, encoder_trains_over_graph_data, encoder_trains_over_hip_batches, route_of
        device = getattr(self, "_device", None)
        inner = self.model.module if hasattr(self.model, "module") else self.model
        em = cfg.preprocessed_metadata.edges[0]
        rank, world = _rank_world()
        if cfg.is_heterogeneous or device is None or device.type != "cuda" or \
                route_of(cfg, self._kwargs) != "hbm" or not (encoder_trains_over_hip_batches(inner.encoder) or
                                                             encoder_trains_over_graph_data(inner.encoder)) or \
                em.positive_edge_info is not None or em.negative_edge_info is not None:
            return None
        # WORLD_SIZE > 1 (round 5; training_process.py:86-119: DDP around the model, the ranks split the batches): trainerArgs
        # hbm_graph = "replica" (default: the whole graph on every rank) | "sharded" (rank r holds the in-edge rows and
        # feature rows of the nodes with id % world == r — configs[4] on a graph larger than one GPU; a batch's remote
        # neighbours and rows arrive through the staged sharded plan, the supervision edges stay a replica)
        want_shards = str(self._kwargs.get("hbm_graph", "replica")).lower() == "sharded" and world > 1
        if any((cfg.dataset_split_uri(sp) and tfrecord_files(cfg.dataset_split_uri(sp))) for sp in ("train", "val", "test")):
            return None
        try:
            res = ResidentGraph(cfg, device, rank=rank, world=world, sharded=want_shards, need_out_graph=True)
        except NotImplementedError:
            return None
        self._resident = res
        self.hbm_graph = "sharded" if res.sharded else "replica"
        res.train_as_graph_data = res.sharded or not encoder_trains_over_hip_batches(inner.encoder)
        # (a GAT encoder reads the stored rows in place where its input-side training forward applies, and gathers the dense
        # matrix itself where it does not: GraphData.features(); on a sharded graph the rows are the pulled ones)
        res.defer_x = (not res.sharded) and type(inner.encoder).__name__ == "GAT" and \
            type(inner.encoder).__module__ == "gigl_amd.models_attn"
        # one engine for the job: the encoder / decoder run on the resident graph's engine
        if self._engine is not None and self._engine is not res.engine:
            self._engine.close()
        self._engine = res.engine
        inner.encoder.engine = inner.decoder.engine = res.engine
        ids, n_pos = res.nablp_anchor_order(cfg.num_positive_samples)
        for sp, want in (("train", range(0, 8)), ("val", (8,)), ("test", (9,))):
            m = np.isin(ids % 10, list(want)) if ids.size >= 100 else np.ones(ids.size, dtype=bool)
            self._hbm_anchors[sp] = (ids[m], n_pos[m])
        return self._hbm_anchors

    def _hbm_typed(self, cfg: GbmlConfigPbWrapper):
        """the typed (heterogeneous) link-prediction job on the in-HBM route: -> {"sampler", "dags", "pos_et", "splits":
        {split: root ids}} when the typed tables are resident in HBM and every training batch is sampled there
        (graphdb_sampler.nablp_batch_graph / batch_graph), else None.  Asked for explicitly (data_route / GIGL_AMD_ROUTE =
        hbm: `auto` keeps typed graphs on TFRecords), one process, no split-generator output; the roots of a split and
        their order are those of the TFRecord route without split files — the sampler's typed main samples (roots of the
        supervision edge type's source type with at least one sampled positive, numMaxTrainingSamplesToOutput applied)
        in file order, root id % 10 (0-7 train, 8 val, 9 test), fewer than 100 samples whole in every split."""
        if getattr(self, "_hbm_typed_state", None) is not None:
            return self._hbm_typed_state or None
        self._hbm_typed_state = {}
        import os
        from .hbm import planned_root_order, route_of
        device = getattr(self, "_device", None)
        if not cfg.is_heterogeneous or device is None or device.type != "cuda" or _rank_world()[1] > 1 or \
                (self._kwargs.get("data_route") or os.environ.get("GIGL_AMD_ROUTE") or "auto").lower() != "hbm":
            return None
        if any((cfg.dataset_split_uri(sp) and tfrecord_files(cfg.dataset_split_uri(sp))) for sp in ("train", "val", "test")):
            return None
        from .graphdb_sampler import EdgeType, HipGraphDBSampler
        from .subgraph_sampler import load_preprocessed_typed_graph, sampling_op_dags
        pos_et = EdgeType(*cfg.supervision_edge_types[0])
        node_types, num, ids, feats, edges, cet, efeats = load_preprocessed_typed_graph(cfg)
        seed = 42 if cfg.permutation_strategy == "deterministic" else 1 + int.from_bytes(os.urandom(3), "little") % ((1 << 20) - 1)
        smp = HipGraphDBSampler(node_types, num, edges, cet, feats, device=device.index or 0, sampling_seed=seed,
                                edge_features=efeats)
        dags = sampling_op_dags(cfg, list(dict.fromkeys([pos_et.src_node_type, pos_et.dst_node_type])))
        # the sampler job's main samples (SubgraphSampler._run_graphdb_nablp): which roots have one, in which order
        roots = np.asarray(ids[pos_et.src_node_type], dtype=np.int64)
        cap = cfg.num_max_training_samples_to_output
        if 0 < cap < roots.size:
            roots = np.sort(np.random.default_rng(seed).choice(roots, size=cap, replace=False))
        if not cfg.should_include_isolated_nodes_in_training:
            keep = []
            for lo in range(0, roots.size, 1 << 16):
                chunk = roots[lo:lo + (1 << 16)]
                pos = smp.sample_positives(chunk, pos_et, cfg.num_positive_samples, dags[pos_et.src_node_type])
                keep.append((pos.nbr.view(chunk.size, -1) != -1).any(dim=1).cpu().numpy())
            roots = roots[np.concatenate(keep)] if keep else roots
        order = planned_root_order(roots, cfg.nablp_tfrecord_uri_prefix)
        splits = {}
        for sp, want in (("train", range(0, 8)), ("val", (8,)), ("test", (9,))):
            m = np.isin(order % 10, list(want)) if order.size >= 100 else np.ones(order.size, dtype=bool)
            splits[sp] = order[m]
        dst_t = pos_et.dst_node_type
        prefix = cfg.random_negative_tfrecord_uri_prefixes.get(dst_t)
        rn_order = planned_root_order(np.asarray(ids[dst_t], dtype=np.int64), prefix) if prefix else np.asarray(ids[dst_t], dtype=np.int64)
        self._hbm_typed_state = {"sampler": smp, "dags": dags, "pos_et": pos_et, "splits": splits, "rn_order": rn_order}
        self._resident = smp  # (what GnnTrainingProcess reports as the route; closed with the spec)
        return self._hbm_typed_state

    def _typed_main_batches_hbm(self, cfg: GbmlConfigPbWrapper, st: dict, split: str, loop: bool):
        from .batches import HeteroNodeAnchorBasedLinkPredictionBatch
        smp, dags, pos_et = st["sampler"], st["dags"], st["pos_et"]
        roots, bs, P = st["splits"][split], self.main_sample_batch_size, cfg.num_positive_samples
        name_to_cnt = {v: k for k, v in cfg.condensed_node_type_map.items()}
        cet = [c for c, triple in cfg.condensed_edge_type_map.items() if tuple(triple) == (pos_et.src_node_type, pos_et.relation, pos_et.dst_node_type)][0]
        spans = [(lo, min(lo + bs, roots.size)) for lo in range(0, roots.size, bs)]
        for lo, hi in (cycle(spans) if (loop and spans) else spans):
            chunk = roots[lo:hi]
            graph, root_index, pos_local, uniq = smp.nablp_batch_graph(chunk, pos_et, P, dags[pos_et.src_node_type],
                                                                       dags[pos_et.dst_node_type])
            smp.engine.synchronize()
            pl = pos_local.cpu()
            yield HeteroNodeAnchorBasedLinkPredictionBatch(
                graph=graph, root_condensed_node_type=name_to_cnt[pos_et.src_node_type], root_node_indices=root_index.cpu(),
                pos_targets={cet: [row[row >= 0] for row in pl]}, hard_neg_targets={},
                condensed_node_type_to_subgraph_id_to_global_node_id={name_to_cnt[t]: u.cpu().numpy().astype(np.int64)
                                                                      for t, u in uniq.items()})

    def _typed_rn_batches_hbm(self, cfg: GbmlConfigPbWrapper, st: dict, batch_size: int):
        from .batches import HeteroRootedNodeNeighborhoodBatch
        smp, dags, dst_t = st["sampler"], st["dags"], st["pos_et"].dst_node_type
        name_to_cnt = {v: k for k, v in cfg.condensed_node_type_map.items()}
        order = st["rn_order"]
        while order.size:
            for lo in range(0, order.size, batch_size):
                chunk = order[lo:lo + batch_size]
                graph, root_index, uniq = smp.batch_graph(chunk, dst_t, dags[dst_t])
                smp.engine.synchronize()
                c = name_to_cnt[dst_t]
                yield HeteroRootedNodeNeighborhoodBatch(
                    graph=graph, condensed_node_type_to_root_node_indices_map={c: root_index.cpu()},
                    root_nodes=[(c, int(v)) for v in chunk.tolist()],
                    condensed_node_type_to_subgraph_id_to_global_node_id={
                        name_to_cnt[t]: dict(enumerate(u.cpu().tolist())) for t, u in uniq.items()})

    def close(self) -> None:
        st = getattr(self, "_hbm_typed_state", None)
        if st:
            st["sampler"].close()
            self._hbm_typed_state = None
            self._resident = None
        if self._resident is not None:
            self._resident.close()  # (its engine is the spec's engine on this route)
            self._resident = None
            self._engine = None
        self._hbm_anchors = None

    # ---- data (dataset/dataloader roles of NodeAnchorBasedLinkPredictionDatasetDataloaders)
    def _main_batches(self, cfg: GbmlConfigPbWrapper, split: str, loop: bool):
        hbm = self._hbm_split(cfg)
        if hbm is not None:
            ids, n_pos = hbm[split]
            yield from self._resident.nablp_batches(ids, n_pos, self.main_sample_batch_size, cfg.num_positive_samples,
                                                    loop=loop)
            return
        rank, world = _rank_world()
        uri = cfg.dataset_split_uri(split)
        if cfg.is_heterogeneous:
            st = self._hbm_typed(cfg)
            if st is not None:
                yield from self._typed_main_batches_hbm(cfg, st, split, loop)
                return
            from .batches import HeteroNodeAnchorBasedLinkPredictionBatch
            files = tfrecord_files(uri) if uri and tfrecord_files(uri) else tfrecord_files(cfg.nablp_tfrecord_uri_prefix)
            raw = [b for chunk in iterate_tfrecord_batches(files, 10 ** 9, rank=rank, world_size=world) for b in chunk]
            if not (uri and tfrecord_files(uri)) and len(raw) >= 100:  # no split-generator output: a root-id split
                want = {"train": range(0, 8), "val": (8,), "test": (9,)}[split]
                raw = [r for r in raw
                       if wire.NodeAnchorBasedLinkPredictionSample.FromString(r).root_node.node_id % 10 in want]
            bs = self.main_sample_batch_size
            chunks = [raw[i:i + bs] for i in range(0, len(raw), bs)]
            for chunk in (cycle(chunks) if (loop and chunks) else chunks):
                yield HeteroNodeAnchorBasedLinkPredictionBatch.process_raw_pyg_samples_and_collate_fn(
                    chunk, cfg.condensed_node_type_map, cfg.condensed_edge_type_map)
            return
        if uri and tfrecord_files(uri):
            raw = [b for chunk in iterate_tfrecord_batches(tfrecord_files(uri), 10 ** 9, rank=rank, world_size=world)
                   for b in chunk]
            bs = self.main_sample_batch_size
            chunks = [raw[i:i + bs] for i in range(0, len(raw), bs)]
            for chunk in (cycle(chunks) if (loop and chunks) else chunks):
                yield NodeAnchorBasedLinkPredictionBatch.process_raw_pyg_samples_and_collate_fn(chunk)
            return
        files = tfrecord_files(cfg.nablp_tfrecord_uri_prefix)
        want = {"train": range(0, 8), "val": (8,), "test": (9,)}[split]
        raw = [b for chunk in iterate_tfrecord_batches(files, 10 ** 9, rank=rank, world_size=world) for b in chunk]
        samples = [wire.NodeAnchorBasedLinkPredictionSample.FromString(b) for b in raw]
        part = samples if len(samples) < 100 else [s for s in samples if s.root_node.node_id % 10 in want]
        bs = self.main_sample_batch_size
        chunks = [part[i:i + bs] for i in range(0, len(part), bs)]
        for chunk in (cycle(chunks) if (loop and chunks) else chunks):
            yield NodeAnchorBasedLinkPredictionBatch.collate_pyg_node_anchor_based_link_prediction_minibatch(chunk)

    def _random_negative_batches(self, cfg: GbmlConfigPbWrapper, batch_size: int, split: str = "train"):
        """always looped, like the reference's LoopyIterableDataset for random negatives"""
        if self._hbm_split(cfg) is not None:
            yield from self._resident.random_negative_batches(batch_size)
            return
        rank, world = _rank_world()
        split_uris = cfg.random_negative_split_uris(split)
        if cfg.is_heterogeneous:  # random negatives of the supervision edge type's destination node type
            st = self._hbm_typed(cfg)
            if st is not None:
                yield from self._typed_rn_batches_hbm(cfg, st, batch_size)
                return
            from .batches import HeteroRootedNodeNeighborhoodBatch
            dst_t = cfg.supervision_edge_types[0][2]
            prefix = split_uris.get(dst_t)
            if not (prefix and tfrecord_files(prefix)):
                prefix = cfg.random_negative_tfrecord_uri_prefixes[dst_t]
            for raw in iterate_tfrecord_batches(tfrecord_files(prefix), batch_size, rank=rank, world_size=world, loop=True):
                yield HeteroRootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(
                    raw, cfg.condensed_node_type_map, cfg.condensed_edge_type_map)
            return
        prefix = next(iter(split_uris.values()), None)
        if not (prefix and tfrecord_files(prefix)):
            prefix = next(iter(cfg.random_negative_tfrecord_uri_prefixes.values()))
        for raw in iterate_tfrecord_batches(tfrecord_files(prefix), batch_size, rank=rank, world_size=world, loop=True):
            yield RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(raw, node_type=cfg.node_types[0])

    # ---- loops
    def _library_train_plan(self, cfg: GbmlConfigPbWrapper):
        """-> engine.NablpTrainPlan when this job's training step can run as ONE library call per batch
        (gigl_nablp_train_plan_*: both encodes, the head, the backward and Adam inside the library, a hipGraph per step),
        else None (the autograd loop below).  That is: the in-HBM route with a whole replica in one process, the plain
        mean-GraphSAGE encoder — or configs[4]'s two-layer GAT (engine.GatNablpTrainPlan.applies) —, the inner-product
        decoder, the Retrieval task alone with its own cross-entropy and no
        candidate-sampling correction, torch.optim.Adam with its default betas / eps, a constant learning rate, no gradient
        clipping; trainerArgs train_plan = "off" keeps the autograd loop."""
        if str(self._kwargs.get("train_plan", "auto")).lower() == "off" or self._hbm_split(cfg) is None:
            return None
        res = self._resident
        inner = self.model.module if hasattr(self.model, "module") else self.model
        enc, dec = inner.encoder, inner.decoder
        tasks = list(self.tasks._task_to_fn_map.values())
        from ._lib import MODE_SPARK_HASH
        from .engine import GatNablpTrainPlan
        sage = type(enc).__module__ == "gigl_amd.models" and type(enc).__name__ == "GraphSAGE" and not res.train_as_graph_data
        gat = GatNablpTrainPlan.applies(enc, res.feat_dim) and len(res.fanouts) == 2  # (configs[4]'s encoder)
        if _rank_world()[1] > 1 or res.sharded or res.mode != MODE_SPARK_HASH or not (sage or gat) or \
                len(tasks) != 1 or type(tasks[0]) is not Retrieval or \
                tasks[0].should_enable_candidate_sampling_correction or tasks[0].loss._loss is not None or \
                self.tasks._task_to_weights_map.get(tasks[0].task_name) != 1.0 or \
                str(getattr(dec, "decoder_type", "inner_product")).split(".")[-1] != "inner_product" or \
                self._optim_cls is not torch.optim.Adam or self.clip_grad_norm > 0 or \
                self._lr_scheduler_cls is not torch.optim.lr_scheduler.ConstantLR or \
                float(self._lr_scheduler_kwargs.get("factor", 1.0)) != 1.0 or int(enc.conv_layers[-1].out_channels) > 512:
            return None
        # the plan's Adam is torch's with default betas / eps and no amsgrad: any other optimiser setting keeps the autograd loop
        if set(self._optim_kwargs) - {"lr", "weight_decay"}:
            return None
        # ... and it trains exactly the tensors its load / store move: the conv layers' weights (+ biases / attention vectors).
        # A model with any other trainable parameter (a decoder MLP, a final linear, batch norm, embeddings) would have it
        # silently left untrained
        covered = {id(p) for c in enc.conv_layers for p in c.parameters()}
        if any(p.requires_grad and id(p) not in covered for p in inner.parameters()):
            return None
        from ._lib import GiglError
        from .engine import NablpTrainPlan
        try:
            return (NablpTrainPlan if sage else GatNablpTrainPlan)(
                res.engine, enc, self.main_sample_batch_size, cfg.num_positive_samples,
                self.random_negative_sample_batch_size, res.fanouts, temperature=float(tasks[0].loss._temperature or 0.0),
                remove_accidental_hits=bool(tasks[0].loss._remove_accidental_hits), lr=self._optim_kwargs["lr"],
                weight_decay=self._optim_kwargs["weight_decay"])
        except (NotImplementedError, GiglError):
            return None

    def _train_with_plan(self, plan, cfg: GbmlConfigPbWrapper, device: torch.device, profiler=None) -> None:
        """train()'s loop with the step inside the library: batches are (roots, positives per anchor) only — the plan
        samples, collates, encodes, scores, backpropagates and updates in one call; the model's parameters are written
        back (plan.store) whenever the Python side needs them (validation, early stopping, the end)"""
        from ._lib import MODE_SPARK_HASH
        res = self._resident
        inner = self.model.module if hasattr(self.model, "module") else self.model
        ids, n_pos = self._hbm_split(cfg)["train"]
        main = res.nablp_root_batches(ids, n_pos, self.main_sample_batch_size, cfg.num_positive_samples)
        rn = res.random_negative_root_batches(self.random_negative_sample_batch_size)
        val_main = self._main_batches(cfg, "val", loop=True)
        val_rn = self._random_negative_batches(cfg, self.random_negative_sample_batch_size_for_evaluation, split="val")
        self.model.train()
        self.train_plan_steps = 0
        every = max(self.validate_every_n_batches, 1)
        def with_next(it):  # (batch, the batch after it | None)
            it = iter(it)
            cur = next(it, None)
            while cur is not None:
                nxt = next(it, None)
                yield cur, nxt
                cur = nxt
        for batch_index, (((roots, cnt, _), rn_roots), nxt) in enumerate(with_next(zip(main, rn)), start=1):
            # the next batch's roots are announced with this step: their sampling + union run beside this step's layers
            # (a batch beyond the plan's workspace — NaN loss, nothing trained — is redone once the plan has grown:
            # NablpTrainPlan.step_checked; a NaN that survives is the loss's own and ends the run like the autograd loop's would)
            loss = plan.step_checked(roots, cnt, rn_roots, sampling_seed=res.seed, mode=MODE_SPARK_HASH,
                                     next_roots=None if nxt is None else (nxt[0][0], nxt[1]))
            self.train_plan_steps += 1
            self.history.append({"batch": batch_index, "loss": loss})
            if loss != loss:
                raise FloatingPointError(f"link-prediction training: the loss of batch {batch_index} is NaN (the plan's workspace "
                                         f"{'was grown' if plan.wide else 'did not overflow'}: the batch itself produced it)")
            if batch_index % every == 0:
                plan.store(inner.encoder)
                metrics = self.validate(val_main, val_rn, cfg, device, self.num_val_batches)
                self.history[-1]["val"] = metrics
                if self.early_stopper.should_early_stop(metrics, self.model):
                    break
                self.model.train()
            if profiler is not None:
                profiler.step()
        plan.store(inner.encoder)
        if not self.early_stopper.best_val_model:
            metrics = self.validate(val_main, val_rn, cfg, device, self.num_val_batches)
            self.early_stopper.should_early_stop(metrics, self.model)
        assert len(self.early_stopper.best_val_model) > 0
        self.model.load_state_dict(self.early_stopper.best_val_model)

    def train(self, gbml_config_pb_wrapper: GbmlConfigPbWrapper, device: torch.device, profiler=None) -> None:
        self._ensure_engine(device)
        cfg = gbml_config_pb_wrapper
        _, world = _rank_world()
        plan = self._library_train_plan(cfg)
        if plan is not None:
            try:
                return self._train_with_plan(plan, cfg, device, profiler)
            finally:
                plan.close()
        main = self._main_batches(cfg, "train", loop=False)
        rn = self._random_negative_batches(cfg, self.random_negative_sample_batch_size)
        val_main = self._main_batches(cfg, "val", loop=True)
        val_rn = self._random_negative_batches(cfg, self.random_negative_sample_batch_size_for_evaluation, split="val")
        self.model.train()
        every = max(self.validate_every_n_batches // world, 1)
        for batch_index, (main_batch, rn_batch) in enumerate(zip(main, rn), start=1):
            self._optimizer.zero_grad()
            task_inputs = infer_task_inputs(self.model, cfg, main_batch, rn_batch, should_eval=False, device=device,
                                            need_batch_scores="batch_scores" in self.tasks.result_types)
            loss, _ = self.tasks.calculate_losses(task_inputs, cfg, should_eval=False, device=device)
            loss.backward()
            if self.clip_grad_norm > 0:
                nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_grad_norm)
            self._optimizer.step()
            self._lr_scheduler.step()
            self.history.append({"batch": batch_index, "loss": float(loss)})
            if batch_index % every == 0:
                if torch.distributed.is_available() and torch.distributed.is_initialized():
                    torch.distributed.barrier()
                metrics = self.validate(val_main, val_rn, cfg, device, self.num_val_batches)
                self.history[-1]["val"] = metrics
                if self.early_stopper.should_early_stop(metrics, self.model):
                    break
            if profiler is not None:
                profiler.step()
        if not self.early_stopper.best_val_model:  # fewer train batches than val_every_num_batches: validate once
            metrics = self.validate(val_main, val_rn, cfg, device, self.num_val_batches)
            self.early_stopper.should_early_stop(metrics, self.model)
        assert len(self.early_stopper.best_val_model) > 0
        self.model.load_state_dict(self.early_stopper.best_val_model)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier()

    @no_grad_eval
    def validate(self, main_data_loader, random_negative_data_loader, gbml_config_pb_wrapper: GbmlConfigPbWrapper,
                 device: torch.device, num_batches: int) -> Dict[EvalMetricType, Any]:
        ks = torch.tensor(KS_FOR_EVAL, dtype=torch.int64, device=device)
        n_rank_nodes = 0
        metrics = {EvalMetricType.mrr: torch.zeros(1, device=device), EvalMetricType.loss: torch.zeros(1, device=device),
                   EvalMetricType.hits: torch.zeros(len(KS_FOR_EVAL), device=device)}
        _, world = _rank_world()
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        per_rank = num_batches // world if distributed else num_batches
        seen = 0
        for batch_idx, (main_batch, rn_batch) in enumerate(zip(main_data_loader, random_negative_data_loader)):
            if batch_idx >= per_rank:
                break
            seen += 1
            ti = infer_task_inputs(self.model, gbml_config_pb_wrapper, main_batch, rn_batch, should_eval=True,
                                   device=device)
            loss, _ = self.tasks.calculate_losses(ti, gbml_config_pb_wrapper, should_eval=True, device=device)
            metrics[EvalMetricType.loss] += loss
            for result in ti.batch_scores:
                for _, bs in result.items():
                    if bs.pos_scores.numel():
                        n_rank_nodes += 1
                        metrics[EvalMetricType.hits] += hit_rate_at_k(bs.pos_scores, bs.random_neg_scores, ks)
                        metrics[EvalMetricType.mrr] += mean_reciprocal_rank(bs.pos_scores, bs.random_neg_scores)
        metrics[EvalMetricType.hits] /= max(n_rank_nodes, 1)
        metrics[EvalMetricType.mrr] /= max(n_rank_nodes, 1)
        metrics[EvalMetricType.loss] /= max(min(per_rank, seen), 1)
        if distributed:
            torch.distributed.barrier()
            for k in metrics:
                torch.distributed.all_reduce(metrics[k], op=torch.distributed.ReduceOp.SUM)
                metrics[k] = metrics[k] / world
        return {k: (v.tolist() if v.shape[0] > 1 else v.item()) for k, v in metrics.items()}

    def eval(self, gbml_config_pb_wrapper: GbmlConfigPbWrapper, device: torch.device) -> EvalMetricsCollection:
        self._ensure_engine(device)
        m = self.validate(self._main_batches(gbml_config_pb_wrapper, "test", loop=False),
                          self._random_negative_batches(gbml_config_pb_wrapper,
                                                        self.random_negative_sample_batch_size_for_evaluation,
                                                        split="test"),
                          gbml_config_pb_wrapper, device, self.num_test_batches)
        hits = [EvalMetric(name=f"HitRate_at_{k}", value=rate) for k, rate in zip(KS_FOR_EVAL, m[EvalMetricType.hits])]
        return EvalMetricsCollection(metrics=[
            EvalMetric.from_eval_metric_type(EvalMetricType.mrr, m[EvalMetricType.mrr]),
            EvalMetric.from_eval_metric_type(EvalMetricType.loss, m[EvalMetricType.loss]), *hits])

    @no_grad_eval
    def infer_batch(self, batch: RootedNodeNeighborhoodBatch, device: torch.device = torch.device("cpu")
                    ) -> InferBatchResults:
        from .hbm import HbmRootBatch
        if isinstance(batch, HbmRootBatch):  # roots of a graph resident in HBM (gigl_amd/hbm.py)
            inner = self.model.module if hasattr(self.model, "module") else self.model
            enc = inner.encoder if hasattr(inner, "encoder") else inner
            return InferBatchResults(embeddings=batch.resident.encode(enc, batch), predictions=None)
        self._ensure_engine(device)
        keys = list(batch.condensed_node_type_to_root_node_indices_map.keys())
        assert len(keys) == 1, ("RootedNodeNeighborhoodBatch for inference must have only one root node type. "
                                f"Found root node types: {keys}")
        idx = batch.condensed_node_type_to_root_node_indices_map[keys[0]].to(device)
        if self._cfg is not None and self._cfg.is_heterogeneous:  # (spec :632-660: the batch's one root node type)
            node_type = self._cfg.condensed_node_type_map[keys[0]]
            out = self.model(batch.graph.to(device), [node_type])[node_type]
            return InferBatchResults(embeddings=out[idx], predictions=None)
        out = self.model(batch.graph.to(device))
        return InferBatchResults(embeddings=out[idx], predictions=None)
