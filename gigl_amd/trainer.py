"""Trainer — drop-in for the reference component

    python -m gigl.src.training.trainer --job_name --task_config_uri --resource_config_uri [--cpu_docker_uri --cuda_docker_uri]
    Trainer().run(applied_task_identifier, task_config_uri, resource_config_uri, cpu_docker_uri=None, cuda_docker_uri=None)
        (python/gigl/src/training/trainer.py:44-51,88-139)

and for the process it launches, GnnTrainingProcess
(python/gigl/src/training/v1/lib/training_process.py:122-139,153-370):
  generate_trainer_instance  import_obj(trainer_cls_path)(**trainer_args), must be a BaseTrainer (:122-139)
  setup_model_device         .to(device) + DistributedDataParallel(broadcast_buffers=False) when WORLD_SIZE>1 (:86-119)
  __run_training             setup_for_training(); train(cfg, device, profiler); save_model on rank 0 (:204-251,59-83)
  __run_model_evaluation     eval(cfg, device) -> metrics file (:172-202)
  should_skip_training       load the pre-trained state_dict and only evaluate (:268-285)
The Vertex-AI launch path (v1/trainer.py:42-78) is out of scope: this always runs in-process on the local GPU(s);
cloud-only arguments are accepted and ignored.  One process per GPU: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*
from the environment (python/gigl/common/utils/torch_training.py:62-74), backend nccl (= RCCL) on GPUs.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from typing import Optional

import torch

from .base import BaseTrainer, EvalMetricsCollection, import_obj
from .config import GbmlConfigPbWrapper


def generate_trainer_instance(cfg: GbmlConfigPbWrapper) -> BaseTrainer:
    cls_path = cfg.trainer_cls_path
    if not cls_path:
        raise ValueError("trainerConfig.trainerClsPath is not set")
    try:
        trainer = import_obj(cls_path)(**cfg.trainer_args)
        assert isinstance(trainer, BaseTrainer)
    except Exception as e:  # logged and re-raised, like the reference (:134-138)
        print(f"Could not instantiate class {cls_path}: {e}", file=sys.stderr)
        raise
    return trainer


def save_model(trainer: BaseTrainer, cfg: GbmlConfigPbWrapper) -> Optional[str]:
    """rank 0 writes the bare state_dict to trainedModelMetadata.trainedModelUri (training_process.py:59-83)"""
    uri = cfg.trained_model_uri
    rank = int(os.environ.get("RANK", "0"))
    if uri is None or rank != 0:
        return None
    model = trainer.model.module if hasattr(trainer.model, "module") else trainer.model
    os.makedirs(os.path.dirname(uri) or ".", exist_ok=True)
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, uri)
    return uri


class GnnTrainingProcess:
    def run(self, task_config_uri: str, device: torch.device, uri_base: Optional[str] = None) -> EvalMetricsCollection:
        import torch.distributed as dist
        world = int(os.environ.get("WORLD_SIZE", "1"))
        started_pg = False
        trainer = None
        if world > 1 and not dist.is_initialized():
            # (GIGL_DIST_BACKEND: e.g. gloo when several ranks share one GPU, where RCCL refuses duplicate devices)
            backend = os.environ.get("GIGL_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
            dist.init_process_group(backend=backend)
            started_pg = True
        try:
            cfg = GbmlConfigPbWrapper.from_uri(task_config_uri, uri_base=uri_base)
            trainer = generate_trainer_instance(cfg)
            state_dict = None
            if cfg.should_skip_training:
                state_dict = torch.load(cfg.trained_model_uri, map_location="cpu")
            trainer.init_model(gbml_config_pb_wrapper=cfg, state_dict=state_dict)
            trainer.model = trainer.model.to(device)
            if world > 1 and trainer.supports_distributed_training:
                trainer.model = torch.nn.parallel.DistributedDataParallel(
                    trainer.model, device_ids=[device.index] if device.type == "cuda" else None,
                    broadcast_buffers=False)
            if not cfg.should_skip_training:
                trainer.setup_for_training()
                trainer.train(gbml_config_pb_wrapper=cfg, device=device, profiler=None)
                save_model(trainer, cfg)
            metrics = trainer.eval(gbml_config_pb_wrapper=cfg, device=device)
            if cfg.eval_metrics_uri and int(os.environ.get("RANK", "0")) == 0:
                os.makedirs(os.path.dirname(cfg.eval_metrics_uri) or ".", exist_ok=True)
                # KFP metrics file shape (training_process.py:154-170)
                json.dump({"metrics": [{"name": m.name, "numberValue": m.value, "format": "RAW"}
                                       for m in metrics.metrics.values()]}, open(cfg.eval_metrics_uri, "w"))
            self.trainer = trainer
            self.route = "hbm" if getattr(trainer, "_resident", None) is not None else "tfrecord"
            return metrics
        finally:
            if trainer is not None and hasattr(trainer, "close"):
                trainer.close()  # (the in-HBM route's resident graph)
            if started_pg:
                dist.destroy_process_group()


class Trainer:
    def run(self, applied_task_identifier: str, task_config_uri: str, resource_config_uri: Optional[str] = None,
            cpu_docker_uri: Optional[str] = None, cuda_docker_uri: Optional[str] = None, *,
            uri_base: Optional[str] = None) -> EvalMetricsCollection:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise RuntimeError("gigl_amd.Trainer needs a HIP device; there is no CPU fallback")
        torch.cuda.set_device(local_rank)
        proc = GnnTrainingProcess()
        metrics = proc.run(task_config_uri, torch.device("cuda", local_rank), uri_base=uri_base)
        self.training_process = proc
        return metrics


def main(argv=None):
    ap = argparse.ArgumentParser(description="MI355X trainer (drop-in for gigl.src.training.trainer)")
    ap.add_argument("--job_name", required=True)
    ap.add_argument("--task_config_uri", required=True)
    ap.add_argument("--resource_config_uri", default=None)
    ap.add_argument("--cpu_docker_uri", default=None)
    ap.add_argument("--cuda_docker_uri", default=None)
    ap.add_argument("--uri_base", default=None)
    a = ap.parse_args(argv)
    print(Trainer().run(a.job_name, a.task_config_uri, a.resource_config_uri, uri_base=a.uri_base))


if __name__ == "__main__":
    main()
