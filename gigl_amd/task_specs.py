"""Task-spec plugin classes (BaseTrainer + BaseInferencer) that run on the HIP path.

HipGraphSageNodeClassificationSpec — the node-classification loop of the reference
  NodeClassificationModelingTaskSpec (python/gigl/src/common/modeling_task_specs/
  node_classification_modeling_task_spec.py:48-299): same constructor kwargs (string-valued
  `trainer_args` / `inferencer_args`: optim_lr 0.01, optim_weight_decay 5e-4, num_epochs 5, out_dim 7,
  main_sample_batch_size 16), same `_train` / `score` / `train` / `eval` / `infer_batch` bodies, with
  `GraphSAGE` wired in place of the stock `TwoLayerGCN` (BASELINE config 1 is
  GraphSAGE node classification; the plugin API allows exactly this substitution).
Data: the Split Generator's train/val/test re-filing of the labeled SupervisedNodeClassificationSample TFRecords
(gigl_amd/split_generator.py writes it; datasetMetadata.supervisedNodeClassificationDataset.*DataUri), read like
the reference's dataloaders.  Without a split-generator output the reference fails on the missing URIs; here the
labeled sampler output is split by root id with a warning (id % 10: 0-7 train, 8 val, 9 test; fixtures of fewer
than 100 samples go whole into every split — metrics are then computed on training data, test use only).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .base import (BaseInferencer, BaseTrainer, EvalMetric, EvalMetricsCollection, EvalMetricType, InferBatchResults,
                   no_grad_eval)
from .batches import SupervisedNodeClassificationBatch, iterate_tfrecord_batches
from .config import GbmlConfigPbWrapper, tfrecord_files
from .models import GraphSAGE
from . import wire


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class HipGraphSageNodeClassificationSpec(BaseTrainer, BaseInferencer):
    def __init__(self, is_training: bool = True, **kwargs) -> None:
        self._optim_lr = float(kwargs.get("optim_lr", 0.01))
        self._optim_weight_decay = float(kwargs.get("optim_weight_decay", 5e-4))
        self._num_epochs = int(kwargs.get("num_epochs", 5))
        self._out_dim = int(kwargs.get("out_dim", 7))
        self._hid_dim = int(kwargs.get("hid_dim", 16))
        self._num_layers = int(kwargs.get("num_layers", 2))
        self._batch_size = int(kwargs.get("main_sample_batch_size", 16))
        # the encoder class is pluggable like the link-prediction spec's (gnn_model_class_path); GAT takes num_heads
        from .base import import_obj
        self._gnn_model = import_obj(str(kwargs.get("gnn_model_class_path", "gigl_amd.models.GraphSAGE")))
        self._encoder_kwargs = {}
        if "num_heads" in kwargs or "heads" in kwargs:
            self._encoder_kwargs["heads"] = int(kwargs.get("num_heads", kwargs.get("heads")))
        if kwargs.get("edge_dim") not in (None, "", "None"):
            self._encoder_kwargs["edge_dim"] = int(kwargs["edge_dim"])
        if "conv" in kwargs:
            self._encoder_kwargs["conv"] = str(kwargs["conv"])
        self._is_training = is_training
        self._model: Optional[torch.nn.Module] = None
        self._engine = None
        self._cfg: Optional[GbmlConfigPbWrapper] = None
        self._kwargs = {k: str(v) for k, v in kwargs.items()}
        self._resident = None  # gigl_amd.hbm.ResidentGraph of the in-HBM route
        self._device: Optional[torch.device] = None

    # ---- BaseModelOperationsProtocol
    @property
    def model(self) -> torch.nn.Module:
        return self._model

    @model.setter
    def model(self, model: torch.nn.Module) -> None:
        self._model = model

    @property
    def supports_distributed_training(self) -> bool:
        return True

    @property
    def supports_hbm_batches(self) -> bool:
        """the in-HBM route (gigl_amd/hbm.py): infer_batch / the training loop also take batches sampled in HBM"""
        from .hbm import encoder_takes_hip_batches, encoder_trains_over_graph_data
        # (encoders without a forward over HipBatches — GIN, GATv2, Transformer ... — get the same in-HBM batch as a
        # GraphData built on the device: hbm.ResidentGraph.encode)
        return self.model is not None and (encoder_takes_hip_batches(self._inner_model()) or
                                           encoder_trains_over_graph_data(self._inner_model()))

    def _inner_model(self) -> torch.nn.Module:
        return self.model.module if hasattr(self.model, "module") else self.model

    def init_model(self, gbml_config_pb_wrapper: GbmlConfigPbWrapper, state_dict=None) -> torch.nn.Module:
        self._cfg = gbml_config_pb_wrapper
        in_dim = gbml_config_pb_wrapper.preprocessed_metadata.nodes[0].feature_dim
        import inspect
        accepted = inspect.signature(self._gnn_model.__init__).parameters
        takes_any = any(p.kind == p.VAR_KEYWORD for p in accepted.values())
        extra = {k: v for k, v in self._encoder_kwargs.items() if k in accepted or takes_any}
        model = self._gnn_model(in_dim=max(in_dim, 1), hid_dim=self._hid_dim, out_dim=self._out_dim,
                                num_layers=self._num_layers, **extra)
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
        inner = self.model.module if hasattr(self.model, "module") else self.model
        inner.engine = self._engine
        return self._engine

    def setup_for_training(self):
        # (capturable: the step counter lives on the device, so the in-HBM route can replay a whole training step as one
        # HIP graph — hbm.GraphedTrainStep; the update itself is the same)
        on_gpu = any(p.is_cuda for p in self.model.parameters())
        self._optimizer = torch.optim.Adam(self.model.parameters(), lr=self._optim_lr,
                                           weight_decay=self._optim_weight_decay, capturable=on_gpu)
        self._train_loss_fn = lambda input, target: F.cross_entropy(input=input, target=target)
        self.model.train()

    # ---- data: the in-HBM route (gigl_amd/hbm.py) — training batches sampled in HBM from the resident graph
    def _hbm_split(self, cfg: GbmlConfigPbWrapper):
        """-> {split: (root ids, labels)} when the job's batches can be sampled in HBM, else None (TFRecord route).
        The roots of a split and their order are those of the files the TFRecord route reads: the labeled samples the
        sampler writes (ascending ids with a label and an in-edge), each sent to the split the split generator's
        node assigner gives its root (TransductiveSupervisedNodeClassificationSplitStrategy, no subsampling); without
        a splitGeneratorConfig: the root-id rule of _split_batches below."""
        from .hbm import ResidentGraph, route_of
        if getattr(self, "_hbm_splits", None) is not None:
            return self._hbm_splits or None
        self._hbm_splits = {}
        route = route_of(cfg, self._kwargs)
        from .hbm import encoder_trains_over_graph_data, encoder_trains_over_hip_batches
        if route != "hbm" or self._device is None or self._device.type != "cuda" or \
                not (encoder_trains_over_hip_batches(self._inner_model()) or encoder_trains_over_graph_data(self._inner_model())):
            return None
        # the same rule as the TFRecord route below: split files written by the SplitGenerator -> its assignment
        # (recomputed from the job's splitGeneratorConfig, no file is read); none written -> the root-id rule
        from .config import _get
        assign = None
        uri = cfg.dataset_split_uri("train")
        if uri and tfrecord_files(uri):
            if not _get(cfg.doc, "datasetConfig.splitGeneratorConfig"):
                return None  # split files of unknown provenance: read them
            from .split_generator import TransductiveSupervisedNodeClassificationSplitStrategy, build_strategy
            try:
                strat = build_strategy(cfg)
            except NotImplementedError:
                return None
            if type(strat) is not TransductiveSupervisedNodeClassificationSplitStrategy or \
                    any(r < 1.0 for r in strat.subsample.ratio.values()):
                return None  # (inductive splits cut the neighbourhoods; subsampling draws per sample: TFRecord route)
            assign = lambda nid: strat.assigner.assign_id(int(nid), 0)
        else:
            import warnings
            warnings.warn("no split-generator output (datasetMetadata.*DataUri is absent or empty): falling back to a "
                          "root-id split of the labeled samples; run the SplitGenerator for the reference's splits",
                          RuntimeWarning, stacklevel=2)
        rank, world = _rank_world()
        # trainerArgs hbm_graph: "replica" (default: every rank holds the whole graph, the ranks only split the batches)
        # | "sharded" (WORLD_SIZE > 1: rank r holds the rows of the nodes with id % world == r; a batch's remote
        # neighbours and feature rows arrive through the sharded plan's exchanges — graphs larger than one GPU's HBM)
        want_shards = str(self._kwargs.get("hbm_graph", "replica")).lower() == "sharded" and world > 1
        self._resident = ResidentGraph(cfg, self._device, rank=rank, world=world, sharded=want_shards)
        self.hbm_graph = "sharded" if self._resident.sharded else "replica"  # (kept after close(): what the job ran on)
        # (plain GraphSAGE trains over HipBatches — and through the library's training plan; every other encoder, and
        # every encoder on a sharded graph, over the same in-HBM batch as a GraphData built on the device)
        self._resident.train_as_graph_data = want_shards or not encoder_trains_over_hip_batches(self._inner_model())
        ids, labels = self._resident.labeled_root_order()
        splits = {}
        if assign is not None:
            which = np.array([assign(i) for i in ids.tolist()], dtype=object)
            for sp in ("train", "val", "test"):
                m = which == sp
                splits[sp] = (ids[m], labels[m])
        else:
            for sp, want in (("train", range(0, 8)), ("val", (8,)), ("test", (9,))):
                m = np.isin(ids % 10, list(want)) if ids.size >= 100 else np.ones(ids.size, dtype=bool)
                splits[sp] = (ids[m], labels[m])
        self._hbm_splits = splits
        return splits

    def close(self) -> None:
        self._graphed_step = None
        if getattr(self, "_train_plan", None) is not None:
            self._train_plan.close()
            self._train_plan = None
        if self._resident is not None:
            self._resident.close()
            self._resident = None
        self._hbm_splits = None

    def _split_batches(self, cfg: GbmlConfigPbWrapper, split: str):
        """in-HBM route: batches of the split's roots sampled in HBM (see _hbm_split).  TFRecord route: the split
        generator's output for `split` (datasetMetadata.supervisedNodeClassificationDataset.*DataUri,
        read like the reference's dataloaders); without one configured / written, the labeled sampler output is
        used directly: root id % 10 (0-7 train, 8 val, 9 test), tiny fixtures (< 100 samples) whole in every split"""
        hbm = self._hbm_split(cfg)
        if hbm is not None:
            ids, labels = hbm[split]
            yield from self._resident.train_batches(ids, labels, self._batch_size)
            return
        rank, world = _rank_world()
        uri = cfg.dataset_split_uri(split)
        if uri and tfrecord_files(uri):
            for raw in iterate_tfrecord_batches(tfrecord_files(uri), self._batch_size, rank=rank, world_size=world):
                yield SupervisedNodeClassificationBatch.process_raw_pyg_samples_and_collate_fn(
                    raw, node_type=cfg.node_types[0])
            return
        import warnings
        warnings.warn(f"no split-generator output for the {split!r} split (datasetMetadata.*DataUri is absent or empty):"
                      " falling back to a root-id split of the labeled sampler output; run the SplitGenerator for the"
                      " reference's splits", RuntimeWarning, stacklevel=2)
        files = tfrecord_files(cfg.labeled_tfrecord_uri_prefix)
        want = {"train": range(0, 8), "val": (8,), "test": (9,)}[split]
        for raw in iterate_tfrecord_batches(files, 10 ** 9, rank=rank, world_size=world):
            samples = [wire.SupervisedNodeClassificationSample.FromString(b) for b in raw]
            part = [s for s in samples if s.root_node.node_id % 10 in want]
            if len(samples) < 100:  # tiny fixture: every split sees everything
                part = samples
            for i in range(0, len(part), self._batch_size):
                yield SupervisedNodeClassificationBatch.collate_pyg_node_classification_minibatch(
                    part[i:i + self._batch_size], node_type=cfg.node_types[0])

    # ---- loops (reference :134-173, :190-227)
    def _train(self, batches, device: torch.device) -> Optional[torch.Tensor]:
        self.model.train()
        loss = None
        for batch in batches:
            self._optimizer.zero_grad()
            inputs = batch.graph.to(device=device)
            root_node_indices = batch.root_node_indices.to(device=device)
            assert batch.root_node_labels is not None, "Labels required for training."
            root_node_labels = batch.root_node_labels.to(device=device)
            out = self.model(inputs)
            loss = self._train_loss_fn(input=out[root_node_indices], target=root_node_labels)
            loss.backward()
            self._optimizer.step()
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        return loss

    def _train_graphed(self, cfg: GbmlConfigPbWrapper, device: torch.device) -> Optional[torch.Tensor]:
        """one epoch of the in-HBM route with every step replayed as one HIP graph (hbm.GraphedTrainStep: the launches of
        _train over batches sampled in HBM, without the host between them).  Single process only (DistributedDataParallel
        hooks its all-reduce into the backward pass); None = not applicable, the caller runs the eager loop."""
        import os
        if os.environ.get("GIGL_AMD_TRAIN_GRAPH", "1") == "0" or _rank_world()[1] > 1 or hasattr(self.model, "module") or \
                getattr(self, "_graph_capture_failed", False):
            return None
        hbm = self._hbm_split(cfg)
        if hbm is None or self._resident.train_as_graph_data:
            return None  # (a GraphData batch reads its sizes on the host: the eager loop)
        ids, labels = hbm["train"]
        if ids.size == 0:
            return None
        b = self._batch_size
        res = self._resident
        loss = self._train_library_plan(res, ids, labels, b, device)
        if loss is not None:
            return loss
        step = getattr(self, "_graphed_step", None)
        if step is None:
            from .hbm import GraphedTrainStep
            r0 = torch.from_numpy(ids[:b].astype(np.uint32).view(np.int32)).to(device)
            try:
                step = GraphedTrainStep(res, self._inner_model(), self._optimizer, b, r0, torch.from_numpy(labels[:b]))
            except Exception as exc:  # noqa: BLE001 — the eager loop is the same step
                import warnings
                warnings.warn(f"training step not captured ({type(exc).__name__}: {exc}); running it eagerly", RuntimeWarning)
                self._graph_capture_failed = True  # (this spec instance only)
                res.engine.bind_stream(torch.cuda.current_stream(device))
                return None
            self._graphed_step = step
        self.model.train()
        roots_all = torch.from_numpy(ids.astype(np.uint32).view(np.int32)).to(device)
        labels_all = torch.from_numpy(labels).to(device)
        loss = None
        for lo in range(0, ids.size, b):
            loss = step.step(roots_all[lo:lo + b], labels_all[lo:lo + b])
        step.stream.synchronize()
        return loss.detach().clone()

    def _train_library_plan(self, res, ids: np.ndarray, labels: np.ndarray, b: int, device: torch.device):
        """one epoch through the library's training plan (engine.SageTrainPlan: sample -> union -> forward -> cross-entropy
        -> backward -> Adam in ONE captured library call per step, no torch kernel in between) — plain mean-GraphSAGE
        encoders under this spec's optimiser settings; None = not applicable (the autograd step is then used).
        GIGL_AMD_TRAIN_PLAN=0 keeps the autograd step."""
        import os
        if os.environ.get("GIGL_AMD_TRAIN_PLAN", "1") == "0":
            return None
        from ._lib import MODE_REPLACE
        from .engine import SageTrainPlan
        model = self._inner_model()
        plan = getattr(self, "_train_plan", None)
        if plan is None:
            if getattr(self, "_train_plan_unavailable", False) or res.mode == MODE_REPLACE:
                return None
            try:
                plan = SageTrainPlan(res.engine, model, b, res.fanouts, lr=self._optim_lr,
                                     weight_decay=self._optim_weight_decay)
            except NotImplementedError:
                self._train_plan_unavailable = True
                return None
            self._train_plan = plan
            self._train_stream = torch.cuda.Stream(device=device)  # (a created stream: the step is replayed as a hipGraph)
        else:
            plan.load(model)  # (the model may have been touched between epochs)
        res.engine.bind_stream(self._train_stream)
        self.model.train()
        roots_all = torch.from_numpy(ids.astype(np.uint32).view(np.int32)).to(device)
        labels_all = torch.from_numpy(labels).to(device)
        torch.cuda.synchronize(device)
        n_steps = -(-ids.size // b)

        def step_args(i):  # (the next batch's sampling + union overlap this batch's layers)
            lo = i * b
            return dict(roots=roots_all[lo:lo + b], labels=labels_all[lo:lo + b], sampling_seed=res.seed, mode=res.mode,
                        next_roots=roots_all[lo + b:lo + 2 * b], next_roots2=roots_all[lo + 2 * b:lo + 3 * b])
        with torch.cuda.stream(self._train_stream):
            # no host read between the steps; a batch beyond the plan's workspace halts the queue on the device and is redone
            # — with everything behind it — once the plan has grown (SageTrainPlan.run_steps)
            losses = plan.run_steps(n_steps, step_args)
        res.engine.synchronize()
        plan.store(model)
        return losses[-1].detach().clone().reshape(())

    @no_grad_eval
    def infer_batch(self, batch: SupervisedNodeClassificationBatch, device: torch.device = torch.device("cpu")
                    ) -> InferBatchResults:
        from .hbm import HbmRootBatch
        if isinstance(batch, HbmRootBatch):  # roots of a graph resident in HBM: sampled, union-ed and encoded there
            embed = batch.resident.encode(self._inner_model(), batch)
            return InferBatchResults(embeddings=embed, predictions=embed.argmax(dim=1))
        self._ensure_engine(device)
        inputs = batch.graph.to(device)
        root_node_indices = batch.root_node_indices.to(device)
        out = self.model(inputs)
        embed = out[root_node_indices]
        pred = embed.argmax(dim=1)
        return InferBatchResults(embeddings=embed, predictions=pred)

    @no_grad_eval
    def score(self, batches, device: torch.device) -> float:
        num_correct = num_evaluated = 0
        for batch in batches:
            assert batch.root_node_labels is not None, "Labels required for scoring."
            results = self.infer_batch(batch=batch, device=device)
            num_correct += int((results.predictions == batch.root_node_labels.to(device)).sum())
            num_evaluated += len(batch.root_node_labels)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
            t = torch.tensor([float(num_correct), float(num_evaluated)], device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            num_correct, num_evaluated = int(t[0].item()), int(t[1].item())
        return num_correct / max(num_evaluated, 1)

    def train(self, gbml_config_pb_wrapper: GbmlConfigPbWrapper, device: torch.device, profiler=None) -> None:
        self._device = device
        self._ensure_engine(device)
        best_val_acc = 0.0
        self.history: List[Dict[str, float]] = []
        for epoch in range(self._num_epochs):
            train_loss = self._train_graphed(gbml_config_pb_wrapper, device)
            if train_loss is None:
                train_loss = self._train(self._split_batches(gbml_config_pb_wrapper, "train"), device)
            val_acc = self.score(self._split_batches(gbml_config_pb_wrapper, "val"), device)
            best_val_acc = max(best_val_acc, val_acc)
            self.history.append({"epoch": epoch, "loss": float(train_loss) if train_loss is not None else float("nan"),
                                 "val_acc": val_acc})
            if profiler is not None:
                profiler.step()

    def eval(self, gbml_config_pb_wrapper: GbmlConfigPbWrapper, device: torch.device) -> EvalMetricsCollection:
        self._device = device
        self._ensure_engine(device)
        test_acc = self.score(self._split_batches(gbml_config_pb_wrapper, "test"), device)
        return EvalMetricsCollection(metrics=[EvalMetric.from_eval_metric_type(EvalMetricType.acc, test_acc)])
