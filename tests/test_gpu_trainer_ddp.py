"""Trainer at world size 2 (training_process.py:86-119: DistributedDataParallel around the spec's model, records
sharded by rank, metrics all-reduced): two processes train the reference's node-classification fixture, the
gradients are averaged by DDP — both ranks end with the SAME weights, which differ from what either rank's half of the
data alone gives — and rank 0 writes the model and the metrics file.
  * two processes sharing the test GPU over gloo (runs on the 1-GPU boxes);
  * two RCCL ranks on two GPUs (self-skips below 2 devices)."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = "configs/snc_frozen_gbml_config.yaml"


def _worker(rank, world, port, backend, workdir, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank if backend == "nccl" else 0),
                          MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GIGL_DIST_BACKEND=backend)
        from gigl_amd.trainer import Trainer
        torch.manual_seed(0)
        tr = Trainer()
        metrics = tr.run("job", CFG, None, uri_base=workdir)
        spec = tr.training_process.trainer
        model = spec.model.module if hasattr(spec.model, "module") else spec.model
        flat = torch.cat([p.detach().float().cpu().reshape(-1) for p in model.parameters()]).numpy()
        q.put((rank, "ok", flat, float(metrics.metrics["acc"].value), [h["loss"] for h in spec.history],
               type(spec.model).__name__))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e), None, None, None))


def _run(world, backend, golden_dir, tmp_path):
    import torch.multiprocessing as mp
    base = tmp_path / "ddp"
    shutil.copytree(os.path.join(golden_dir, "configs"), base / "configs")
    shutil.copytree(os.path.join(golden_dir, "ref_assets"), base / "ref_assets")
    from gigl_amd.subgraph_sampler import SubgraphSampler
    SubgraphSampler().run("job", CFG, None, uri_base=str(base))
    # the reference shards FILES across ranks (data_loaders/utils.py:23-56): two part files -> each rank its own half
    from gigl_amd import wire
    from gigl_amd.config import GbmlConfigPbWrapper, tfrecord_files
    cfg = GbmlConfigPbWrapper.from_uri(CFG, uri_base=str(base))
    files = tfrecord_files(cfg.labeled_tfrecord_uri_prefix)
    recs = [r for f in files for r in wire.read_tfrecords(f)]
    assert len(recs) == 14
    for f in files:
        os.remove(f)
    d = os.path.dirname(files[0])
    wire.write_tfrecords(os.path.join(d, "part-00000.tfrecord"), recs[:7])
    wire.write_tfrecords(os.path.join(d, "part-00001.tfrecord"), recs[7:])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, str(base), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    for rank, status, info, *_ in res:
        assert status == "ok", f"rank {rank}: {info}"
    (_, _, w0, acc0, loss0, kind0), (_, _, w1, acc1, loss1, kind1) = res
    assert kind0 == kind1 == "DistributedDataParallel"
    np.testing.assert_array_equal(w0, w1)          # one model: DDP averaged the gradients of the two halves
    assert acc0 == acc1                            # metrics are all-reduced
    assert all(np.isfinite(loss0)) and all(np.isfinite(loss1)) and loss0 != loss1  # each rank saw its own records
    sd = torch.load(cfg.trained_model_uri, map_location="cpu")  # written by rank 0, without the DDP prefix
    assert "conv_layers.0.lin_l.weight" in sd
    np.testing.assert_array_equal(torch.cat([v.float().reshape(-1) for v in sd.values()]).numpy().sum(), w0.sum())
    assert json.load(open(cfg.eval_metrics_uri))["metrics"][0]["name"] == "acc"


def test_two_ranks_on_one_gpu_over_gloo(golden_dir, tmp_path):
    _run(2, "gloo", golden_dir, tmp_path)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the build boxes have one)")
def test_two_rccl_ranks_on_two_gpus(golden_dir, tmp_path):
    _run(2, "nccl", golden_dir, tmp_path)


def _worker_cfg(rank, world, port, workdir, cfg_uri, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port), GIGL_DIST_BACKEND="gloo")
        from gigl_amd.trainer import Trainer
        torch.manual_seed(0)
        tr = Trainer()
        metrics = tr.run("job", cfg_uri, None, uri_base=workdir)
        spec = tr.training_process.trainer
        model = spec.model.module if hasattr(spec.model, "module") else spec.model
        flat = torch.cat([p.detach().float().cpu().reshape(-1) for p in model.parameters()]).numpy()
        q.put((rank, "ok", flat, float(metrics.metrics["acc"].value), [h["loss"] for h in spec.history],
               getattr(spec, "hbm_graph", None) == "sharded"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e), None, None, None))


@pytest.mark.parametrize("encoder", ["gigl_amd.models.GraphSAGE", "gigl_amd.models_more.GIN"])
def test_trainer_over_a_hash_partitioned_graph_matches_the_replica_route(golden_dir, tmp_path, encoder):
    """trainerArgs hbm_graph = sharded at world size 2 (two processes on the test GPU, gloo): rank r holds the rows of
    the nodes with id % 2 == r only, a batch's remote neighbours and feature rows arrive through the sharded plan's
    exchanges (a STAGED plan: hbm.ResidentGraph.graph_data), the encoder trains over the batch graph built on the
    device, DDP averages the gradients — the same batches, hence the same loss history, weights and metrics as with a
    whole replica of the graph on every rank (up to the summation order of the backward's atomics)"""
    import torch.multiprocessing as mp
    import yaml
    base = tmp_path / "ddp_sharded"
    shutil.copytree(os.path.join(golden_dir, "configs"), base / "configs")
    shutil.copytree(os.path.join(golden_dir, "ref_assets"), base / "ref_assets")
    doc = yaml.safe_load(open(base / CFG))
    runs = {}
    for k, mode in enumerate(("replica", "sharded")):
        args = dict(doc["trainerConfig"].get("trainerArgs") or {})
        args.update(hbm_graph=mode, data_route="hbm", gnn_model_class_path=encoder)
        doc["trainerConfig"]["trainerArgs"] = args
        doc["sharedConfig"]["trainedModelMetadata"] = {"trainedModelUri": f"out/ddp_{mode}/model.pt",
                                                       "evalMetricsUri": f"out/ddp_{mode}/eval_metrics.json"}
        uri = f"configs/snc_{mode}_gbml_config.yaml"
        yaml.safe_dump(doc, open(base / uri, "w"))
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 29850 + (os.getpid() + 7 * k) % 40
        procs = [ctx.Process(target=_worker_cfg, args=(r, 2, port, str(base), uri, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        for rank, status, info, *_ in res:
            assert status == "ok", f"{mode} rank {rank}: {info}"
        runs[mode] = res
    for r in range(2):
        assert runs["sharded"][r][5] is True and runs["replica"][r][5] is False
        h_s, h_r = runs["sharded"][r][4], runs["replica"][r][4]
        assert len(h_s) == len(h_r) >= 1
        np.testing.assert_allclose(h_s, h_r, rtol=5e-3)
        np.testing.assert_allclose(runs["sharded"][r][2], runs["replica"][r][2], rtol=2e-2, atol=max(0.02, 0.01 * len(h_s)))
        assert abs(runs["sharded"][r][3] - runs["replica"][r][3]) <= 0.15
    np.testing.assert_array_equal(runs["sharded"][0][2], runs["sharded"][1][2])  # one model on both ranks


NABLP_CFG = "configs/nablp_frozen_gbml_config.yaml"


def _worker_nablp(rank, world, port, workdir, cfg_uri, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port), GIGL_DIST_BACKEND="gloo", GIGL_AMD_ROUTE="hbm")
        from gigl_amd.trainer import Trainer
        torch.manual_seed(0)
        np.random.seed(0)
        tr = Trainer()
        metrics = tr.run("job", cfg_uri, None, uri_base=workdir)
        spec = tr.training_process.trainer
        model = spec.model.module if hasattr(spec.model, "module") else spec.model
        flat = torch.cat([p.detach().float().cpu().reshape(-1) for p in model.parameters()]).numpy()
        q.put((rank, "ok", flat, {k: float(m.value) for k, m in metrics.metrics.items()}, [h["loss"] for h in spec.history],
               getattr(spec, "hbm_graph", None), tr.training_process.route))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e), None, None, None, None))


@pytest.mark.parametrize("encoder", [None, "gigl_amd.models_attn.GAT"])
def test_link_prediction_trainer_over_a_hash_partitioned_graph_matches_the_replica_route(golden_dir, tmp_path, encoder):
    """the link-prediction trainer (BASELINE configs[4]'s task) at world size 2 on the in-HBM route, two processes on the
    test GPU over gloo: trainerArgs hbm_graph = sharded — rank r holds only the in-edge rows and feature rows of the nodes
    with id % 2 == r; main batches (anchors + positives drawn from the replicated supervision edges) and random-negative
    batches are assembled through the STAGED sharded plan's exchanges, DDP averages the gradients — against hbm_graph =
    replica (the whole graph on every rank, the same batches per rank): same loss history, weights and test loss
    (training_process.py:86-119; node_anchor_based_link_prediction_modeling_task_spec.py:334-451)"""
    import torch.multiprocessing as mp
    import yaml
    base = tmp_path / "ddp_lp"
    shutil.copytree(os.path.join(golden_dir, "configs"), base / "configs")
    shutil.copytree(os.path.join(golden_dir, "ref_assets"), base / "ref_assets")
    from gigl_amd.subgraph_sampler import SubgraphSampler
    SubgraphSampler().run("job", NABLP_CFG, None, uri_base=str(base))  # (the random-negative / main file ORDER is read from here)
    shutil.rmtree(os.path.join(str(base), "out", "nablp", "split"), ignore_errors=True)
    doc = yaml.safe_load(open(base / NABLP_CFG))
    runs = {}
    for k, mode in enumerate(("replica", "sharded")):
        args = dict(doc["trainerConfig"].get("trainerArgs") or {})
        args.update(hbm_graph=mode, data_route="hbm")
        if encoder is not None:
            args.update(gnn_model_class_path=encoder, hidden_dim="8", out_channels="8", num_heads="2")
        doc["trainerConfig"]["trainerArgs"] = args
        doc["sharedConfig"]["trainedModelMetadata"] = {"trainedModelUri": f"out/ddp_lp_{mode}/model.pt",
                                                       "evalMetricsUri": f"out/ddp_lp_{mode}/eval_metrics.json"}
        uri = f"configs/nablp_{mode}_gbml_config.yaml"
        yaml.safe_dump(doc, open(base / uri, "w"))
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 29750 + (os.getpid() + 7 * k) % 40
        procs = [ctx.Process(target=_worker_nablp, args=(r, 2, port, str(base), uri, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted((q.get(timeout=900) for _ in range(2)), key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        for rank, status, info, *_ in res:
            assert status == "ok", f"{mode} rank {rank}: {info}"
        runs[mode] = res
    tol = 1e-4 if encoder is None else 5e-3
    for r in range(2):
        assert runs["sharded"][r][5] == "sharded" and runs["replica"][r][5] == "replica"
        assert runs["sharded"][r][6] == runs["replica"][r][6] == "hbm"
        h_s, h_r = runs["sharded"][r][4], runs["replica"][r][4]
        assert len(h_s) == len(h_r) >= 1 and all(np.isfinite(h_s))
        np.testing.assert_allclose(h_s, h_r, rtol=tol)
        np.testing.assert_allclose(runs["sharded"][r][2], runs["replica"][r][2], rtol=1e-3 if encoder is None else 5e-2,
                                   atol=1e-5 if encoder is None else 0.05)
        np.testing.assert_allclose(runs["sharded"][r][3]["loss"], runs["replica"][r][3]["loss"], rtol=tol * 5, atol=1e-6)
    np.testing.assert_array_equal(runs["sharded"][0][2], runs["sharded"][1][2])  # one model on both ranks
