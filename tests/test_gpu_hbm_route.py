"""The in-HBM route behind the drop-in entry points (gigl_amd/hbm.py): Inferencer.run / Trainer.run over a graph and
feature table resident in HBM, batches sampled there and encoded by the one-call plans — against the TFRecord route
(the reference's dataflow: sampler part files -> collate -> infer_batch) and against the fp32 CPU restatement.

Both routes walk the same roots in the same batches, so their rows agree to fp32 rounding (1e-5); a root's embedding
depends on its batch (every layer runs over the batch's union graph), which is what makes the order part of parity."""
import json
import os
import shutil

import numpy as np
import pytest
import torch
import yaml

from gigl_amd import wire
from gigl_amd.config import GbmlConfigPbWrapper, tfrecord_files

SNC = "configs/snc_frozen_gbml_config.yaml"
NABLP = "configs/nablp_frozen_gbml_config.yaml"


def _rows(path):
    return [json.loads(l) for l in open(path)]


def test_planned_root_order_is_the_file_readers_order(tmp_path, monkeypatch):
    """CPU: the computed root order == the order iterate_tfrecord_batches reads the part files _PartWriter writes"""
    from gigl_amd import config
    from gigl_amd.batches import iterate_tfrecord_batches
    from gigl_amd.hbm import planned_root_order
    from gigl_amd.subgraph_sampler import _PartWriter
    monkeypatch.setattr(config, "RECORDS_PER_PART_FILE", 7)
    for n, prefix in ((0, "a/samples/"), (5, "b/samples/"), (7, "c/samples/"), (100, "d/samples/"), (53, "e/pre_")):
        ids = np.arange(3, 3 + 2 * n, 2, dtype=np.int64)
        w = _PartWriter(str(tmp_path / prefix))
        for i in ids.tolist():
            w.add_frame(wire.tfrecord_frame(int(i).to_bytes(8, "little")))
        files = w.close()
        assert sorted(files) == tfrecord_files(str(tmp_path / prefix))
        read = [int.from_bytes(r, "little") for raw in iterate_tfrecord_batches(files, 10) for r in raw]
        assert read == planned_root_order(ids, str(tmp_path / prefix)).tolist()
        if n == 100:
            assert read != ids.tolist()  # (15 files: the permutation moves them)


@pytest.fixture(scope="module")
def workdir(golden_dir, tmp_path_factory):
    base = tmp_path_factory.mktemp("gigl_hbm")
    shutil.copytree(os.path.join(golden_dir, "configs"), base / "configs")
    shutil.copytree(os.path.join(golden_dir, "ref_assets"), base / "ref_assets")
    return str(base)


def _variant(workdir, cfg_uri, name, **inference_paths):
    """the config with its inference outputs re-pointed (one run per route must not overwrite the other's rows)"""
    doc = yaml.safe_load(open(os.path.join(workdir, cfg_uri)))
    info = doc["sharedConfig"]["inferenceMetadata"]["nodeTypeToInferencerOutputInfoMap"]
    for v in info.values():
        for k in list(v):
            v[k] = inference_paths.get(k, v[k].replace("inference/", f"inference_{name}/"))
    out = cfg_uri.replace(".yaml", f"_{name}.yaml")
    yaml.safe_dump(doc, open(os.path.join(workdir, out), "w"))
    return out


@pytest.mark.gpu
def test_inferencer_routes_agree_on_the_reference_fixture(workdir):
    """reference fixture config (16 nodes, fanout [3,3], inferenceBatchSize 8): in-HBM rows == TFRecord-route rows ==
    fp32 CPU forward over the collated batches"""
    from gigl_amd.batches import RootedNodeNeighborhoodBatch, iterate_tfrecord_batches
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd.trainer import Trainer
    from oracle import gnn_ref
    SubgraphSampler().run("job", SNC, None, uri_base=workdir)
    torch.manual_seed(1)
    Trainer().run("job", SNC, None, uri_base=workdir)
    a, b = Inferencer(), Inferencer()
    out_t = a.run("job", _variant(workdir, SNC, "tf"), None, uri_base=workdir, route="tfrecord")
    out_h = b.run("job", _variant(workdir, SNC, "hbm"), None, uri_base=workdir, route="hbm")
    assert a.route == "tfrecord" and b.route == "hbm" and a.rows_written == b.rows_written == 16
    rt, rh = _rows(out_t["embeddings"]), _rows(out_h["embeddings"])
    assert [r["node_id"] for r in rt] == [r["node_id"] for r in rh]  # same roots in the same order
    np.testing.assert_allclose(np.array([r["emb"] for r in rh], np.float32), np.array([r["emb"] for r in rt], np.float32),
                               rtol=1e-5, atol=1e-5)
    assert _rows(out_t["predictions"]) == _rows(out_h["predictions"])
    cfg = GbmlConfigPbWrapper.from_uri(SNC, uri_base=workdir)
    sd = torch.load(cfg.trained_model_uri, map_location="cpu")
    want = {}
    for raw in iterate_tfrecord_batches(tfrecord_files(cfg.unlabeled_tfrecord_uri_prefix), 8):
        bt = RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(raw)
        o = gnn_ref.graphsage_forward(bt.graph.x, bt.graph.edge_index, sd, 2)
        for r, i in zip(bt.root_nodes, bt.condensed_node_type_to_root_node_indices_map[0].tolist()):
            want[r.id] = o[i].numpy()
    for row in rh:
        np.testing.assert_allclose(np.array(row["emb"], np.float32), want[row["node_id"]], rtol=1e-5, atol=1e-5)
    # default route: auto = in-HBM (the tables are readable)
    c = Inferencer()
    c.run("job", _variant(workdir, SNC, "auto"), None, uri_base=workdir)
    assert c.route == "hbm"


@pytest.mark.gpu
def test_inferencer_avro_shards_from_the_hbm_route(workdir):
    """embeddingsPath naming a directory: the in-HBM route hands device rows to the device Avro encoder"""
    from gigl_amd.inferencer import Inferencer
    from oracle import avro
    cfg = GbmlConfigPbWrapper.from_uri(SNC, uri_base=workdir)
    if not os.path.exists(cfg.trained_model_uri):
        pytest.skip("needs the model of the test above")
    out_j = Inferencer().run("job", _variant(workdir, SNC, "hbmj"), None, uri_base=workdir, route="hbm")
    out_a = Inferencer().run("job", _variant(workdir, SNC, "hbma", embeddingsPath="out/snc/inference_hbma/avro/"), None,
                             uri_base=workdir, route="hbm")
    shards = sorted(os.listdir(out_a["embeddings"]))
    assert shards == ["shard_00000000.avro"]
    _, recs = avro.read_embedding_file(open(os.path.join(out_a["embeddings"], shards[0]), "rb").read())
    rows = _rows(out_j["embeddings"])
    assert [r["node_id"] for r in recs] == [r["node_id"] for r in rows]
    for x, y in zip(recs, rows):
        np.testing.assert_array_equal(np.array(x["emb"], np.float32), np.array(y["emb"], np.float32))


@pytest.mark.gpu
def test_link_prediction_inferencer_routes_agree(workdir):
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd.trainer import Trainer
    SubgraphSampler().run("job", NABLP, None, uri_base=workdir)
    torch.manual_seed(1)
    Trainer().run("job", NABLP, None, uri_base=workdir)
    a, b = Inferencer(), Inferencer()
    out_t = a.run("job", _variant(workdir, NABLP, "tf"), None, uri_base=workdir, route="tfrecord")
    out_h = b.run("job", _variant(workdir, NABLP, "hbm"), None, uri_base=workdir, route="hbm")
    assert b.route == "hbm" and a.rows_written == b.rows_written == 27
    rt, rh = _rows(out_t["embeddings"]), _rows(out_h["embeddings"])
    assert [r["node_id"] for r in rt] == [r["node_id"] for r in rh]
    np.testing.assert_allclose(np.array([r["emb"] for r in rh], np.float32), np.array([r["emb"] for r in rt], np.float32),
                               rtol=1e-5, atol=1e-5)


# ---- a products-shaped small graph written as the Data Preprocessor's tables --------------------------------------

def _write_small_job(base, n=20_000, e=150_000, d=32, hid=64, out_dim=16, fan=(10, 5), batch=512, seed=7,
                     directed=False):
    from helpers import rmat_edges
    rng = np.random.default_rng(seed)
    src, dst = rmat_edges(15, e, seed)
    src, dst = (src.astype(np.int64) * 0x9E3779B1) % n, (dst.astype(np.int64) * 0x9E3779B1) % n
    keep = src != dst
    src, dst = src[keep], dst[keep]
    x = rng.standard_normal((n, d)).astype(np.float32)
    os.makedirs(os.path.join(base, "tables/nodes"), exist_ok=True)
    os.makedirs(os.path.join(base, "tables/edges"), exist_ok=True)
    wire.write_tfrecords(os.path.join(base, "tables/nodes/data.tfrecord"), [
        wire.encode_tf_example({"node_id": np.array([i], np.int64), "feat": x[i],
                                "node_label": np.array([i % out_dim], np.int64)}) for i in range(n)])
    wire.write_tfrecords(os.path.join(base, "tables/edges/data.tfrecord"), [
        wire.encode_tf_example({"src": np.array([s], np.int64), "dst": np.array([t], np.int64)})
        for s, t in zip(src.tolist(), dst.tolist())])
    os.makedirs(os.path.join(base, "configs"), exist_ok=True)
    yaml.safe_dump({
        "condensedEdgeTypeToPreprocessedMetadata": {"0": {"dstNodeIdKey": "dst", "srcNodeIdKey": "src", "mainEdgeInfo": {
            "tfrecordUriPrefix": "tables/edges", "featureDim": 0}}},
        "condensedNodeTypeToPreprocessedMetadata": {"0": {"featureDim": d, "featureKeys": ["feat"], "labelKeys": [
            "node_label"], "nodeIdKey": "node_id", "tfrecordUriPrefix": "tables/nodes"}}},
        open(os.path.join(base, "configs/pm.yaml"), "w"))
    spec = "gigl_amd.task_specs.HipGraphSageNodeClassificationSpec"
    args = {"out_dim": str(out_dim), "hid_dim": str(hid), "main_sample_batch_size": "256", "num_epochs": "2"}
    yaml.safe_dump({
        "graphMetadata": {"edgeTypes": [{"dstNodeType": "paper", "relation": "cites", "srcNodeType": "paper"}],
                          "nodeTypes": ["paper"]},
        "taskMetadata": {"nodeBasedTaskMetadata": {"supervisionNodeTypes": ["paper"]}},
        "datasetConfig": {"subgraphSamplerConfig": {
            "numHops": len(fan), "numNeighborsToSample": fan[0], "experimentalFlags": {"permutation_strategy": "deterministic"},
            "subgraphSamplingStrategy": {"messagePassingPaths": {"paths": [{"rootNodeType": "paper", "samplingOps": [
                {"opName": f"hop{k}", "edgeType": {"srcNodeType": "paper", "relation": "cites", "dstNodeType": "paper"},
                 "randomUniform": {"numNodesToSample": f}, "inputOpNames": ([f"hop{k - 1}"] if k else [])}
                for k, f in enumerate(fan)]}]}}}},
        "sharedConfig": {
            "isGraphDirected": bool(directed),
            "flattenedGraphMetadata": {"supervisedNodeClassificationOutput": {
                "labeledTfrecordUriPrefix": "out/labeled/samples/", "unlabeledTfrecordUriPrefix": "out/unlabeled/samples/"}},
            "preprocessedMetadataUri": "configs/pm.yaml",
            "trainedModelMetadata": {"trainedModelUri": "out/model/model.pt", "evalMetricsUri": "out/model/eval.json"},
            "inferenceMetadata": {"nodeTypeToInferencerOutputInfoMap": {"paper": {
                "embeddingsPath": "out/inference/embeddings.jsonl", "predictionsPath": "out/inference/predictions.jsonl"}}}},
        "trainerConfig": {"trainerClsPath": spec, "trainerArgs": args},
        "inferencerConfig": {"inferencerClsPath": spec, "inferencerArgs": {k: args[k] for k in ("out_dim", "hid_dim")},
                             "inferenceBatchSize": batch}},
        open(os.path.join(base, "configs/job.yaml"), "w"))
    return n, src, dst, x


@pytest.fixture(scope="module")
def small_job(tmp_path_factory):
    from gigl_amd.models import GraphSAGE
    base = str(tmp_path_factory.mktemp("gigl_hbm_small"))
    n, src, dst, x = _write_small_job(base)
    torch.manual_seed(3)
    model = GraphSAGE(32, 64, 16, num_layers=2)
    os.makedirs(os.path.join(base, "out/model"), exist_ok=True)
    torch.save(model.state_dict(), os.path.join(base, "out/model/model.pt"))
    return base, n, src, dst, x


@pytest.mark.gpu
def test_inferencer_routes_agree_on_a_products_shaped_small_graph(small_job, monkeypatch):
    """20k nodes, fanout [10,5], inferenceBatchSize 512, part files of 3000 records (7 files: the reader's permutation
    matters), several batches per library call: in-HBM rows == TFRecord-route rows (1e-5), and == the fp32 CPU forward
    over oracle-collated batches for the first and the last (partial) batch"""
    import oracle
    from gigl_amd import config
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from oracle import gnn_ref
    base, n, src, dst, x = small_job
    monkeypatch.setattr(config, "RECORDS_PER_PART_FILE", 3000)
    SubgraphSampler().run("job", "configs/job.yaml", None, uri_base=base)
    cfg = GbmlConfigPbWrapper.from_uri("configs/job.yaml", uri_base=base)
    assert len(tfrecord_files(cfg.unlabeled_tfrecord_uri_prefix)) == 7
    a, b = Inferencer(), Inferencer()
    out_t = a.run("job", _variant(base, "configs/job.yaml", "tf"), None, uri_base=base, route="tfrecord")
    out_h = b.run("job", _variant(base, "configs/job.yaml", "hbm"), None, uri_base=base, route="hbm")
    assert a.rows_written == b.rows_written == n and b.hbm_groups > 1
    rt, rh = _rows(out_t["embeddings"]), _rows(out_h["embeddings"])
    ids = [r["node_id"] for r in rh]
    assert ids == [r["node_id"] for r in rt] and sorted(ids) == list(range(n)) and ids != list(range(n))
    eh, et = np.array([r["emb"] for r in rh], np.float32), np.array([r["emb"] for r in rt], np.float32)
    np.testing.assert_allclose(eh, et, rtol=1e-5, atol=1e-5)
    assert _rows(out_t["predictions"]) == _rows(out_h["predictions"]) or \
        np.mean([p["pred"] == q["pred"] for p, q in zip(_rows(out_t["predictions"]), _rows(out_h["predictions"]))]) > 0.999
    # oracle: sample -> collate -> fp32 forward of the first and of the last (partial: 20000 % 512 = 32 roots) batch
    rowptr, col = oracle.build_csc(n, src.astype(np.uint32), dst.astype(np.uint32), is_directed=False)
    sd = torch.load(cfg.trained_model_uri, map_location="cpu")
    for lo, hi in ((0, 512), (n - n % 512, n)):
        roots = np.array(ids[lo:hi], dtype=np.uint32)
        nbr, _ = oracle.sample_khop(rowptr, col, roots, [10, 5], canonical=True)
        u = oracle.union_build(roots, [10, 5], nbr)
        ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
        o = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"].astype(np.int64)]), ei, sd, 2)
        np.testing.assert_allclose(eh[lo:hi], o[u["root_local"]].numpy(), rtol=1e-5, atol=1e-5)


def _infer_worker(rank, world, port, base, cfg_uri, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port), GIGL_DIST_BACKEND="gloo")
        from gigl_amd import config
        from gigl_amd.inferencer import Inferencer
        config.RECORDS_PER_PART_FILE = 3000
        inf = Inferencer()
        out = inf.run("job", cfg_uri, None, uri_base=base, route="hbm")
        q.put((rank, "ok", out, inf.rows_written))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e), 0))


@pytest.mark.gpu
def test_two_rank_inferencer_with_an_encoder_outside_the_sharded_plan(small_job, monkeypatch):
    """WORLD_SIZE = 2 with a GIN encoder (no one-call sharded plan): every batch is assembled from the two shards by the
    STAGED sharded plan (ResidentGraph.graph_data: union graph + dense feature rows in the rank's HBM) and the encoder's
    own forward runs over it — rows == the single-process in-HBM rows (1e-5)"""
    import torch.multiprocessing as mp
    from gigl_amd import config
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.models_more import GIN
    base, n, *_ = small_job
    monkeypatch.setattr(config, "RECORDS_PER_PART_FILE", 3000)
    doc = yaml.safe_load(open(os.path.join(base, "configs/job.yaml")))
    args = dict(doc["inferencerConfig"].get("inferencerArgs") or {})
    args["gnn_model_class_path"] = "gigl_amd.models_more.GIN"
    doc["inferencerConfig"]["inferencerArgs"] = args
    doc["trainerConfig"]["trainerArgs"] = dict(doc["trainerConfig"].get("trainerArgs") or {}, gnn_model_class_path="gigl_amd.models_more.GIN")
    doc["sharedConfig"]["trainedModelMetadata"]["trainedModelUri"] = "out/model_gin/model.pt"
    yaml.safe_dump(doc, open(os.path.join(base, "configs/job_gin.yaml"), "w"))
    cfg = GbmlConfigPbWrapper.from_uri("configs/job_gin.yaml", uri_base=base)
    from gigl_amd.task_specs import HipGraphSageNodeClassificationSpec
    spec = HipGraphSageNodeClassificationSpec(**cfg.inferencer_args)
    torch.manual_seed(4)
    model = spec.init_model(cfg)
    assert isinstance(model, GIN)
    os.makedirs(os.path.join(base, "out/model_gin"), exist_ok=True)
    torch.save(model.state_dict(), os.path.join(base, "out/model_gin/model.pt"))
    single = Inferencer().run("job", _variant(base, "configs/job_gin.yaml", "g1"), None, uri_base=base, route="hbm")
    want = {r["node_id"]: np.array(r["emb"], np.float32) for r in _rows(single["embeddings"])}
    cfg2 = _variant(base, "configs/job_gin.yaml", "g2")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29760 + os.getpid() % 40
    procs = [ctx.Process(target=_infer_worker, args=(r, 2, port, base, cfg2, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    for rank, status, info, _ in res:
        assert status == "ok", f"rank {rank}: {info}"
    got = {}
    for rank, _, out, n_rows in res:
        for r in _rows(out["embeddings"]):
            assert r["node_id"] not in got
            got[r["node_id"]] = np.array(r["emb"], np.float32)
    assert sorted(got) == sorted(want)
    ids = sorted(want)
    np.testing.assert_allclose(np.stack([got[i] for i in ids]), np.stack([want[i] for i in ids]), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_two_rank_inferencer_over_the_sharded_plan(small_job, monkeypatch):
    """WORLD_SIZE = 2 (two processes sharing the test GPU, gloo + the callback transport): every rank ingests its
    shard (gigl_graph_build_shard_from_coo), batches go to rank c % 2 and run through the sharded plan; the union of the
    ranks' rows == the single-process in-HBM rows (same batches, 1e-5)"""
    import torch.multiprocessing as mp
    from gigl_amd import config
    from gigl_amd.inferencer import Inferencer
    base, n, *_ = small_job
    monkeypatch.setattr(config, "RECORDS_PER_PART_FILE", 3000)
    single = Inferencer().run("job", _variant(base, "configs/job.yaml", "w1"), None, uri_base=base, route="hbm")
    want = {r["node_id"]: np.array(r["emb"], np.float32) for r in _rows(single["embeddings"])}
    cfg2 = _variant(base, "configs/job.yaml", "w2")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 40
    procs = [ctx.Process(target=_infer_worker, args=(r, 2, port, base, cfg2, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    for rank, status, info, _ in res:
        assert status == "ok", f"rank {rank}: {info}"
    got = {}
    for rank, _, out, n_rows in res:
        assert out["embeddings"].endswith(f".rank{rank}")
        rows = _rows(out["embeddings"])
        assert len(rows) == n_rows
        for r in rows:
            assert r["node_id"] not in got
            got[r["node_id"]] = np.array(r["emb"], np.float32)
    assert sorted(got) == sorted(want) and abs(res[0][3] - res[1][3]) <= 512
    ids = sorted(want)
    np.testing.assert_allclose(np.stack([got[i] for i in ids]), np.stack([want[i] for i in ids]), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_trainer_routes_agree(workdir):
    """Trainer.run with training batches sampled in HBM (HipBatch, autograd over the union graph) == Trainer.run over
    the TFRecord route: same initialisation -> the same loss history and the same trained weights (fp32 rounding)"""
    from gigl_amd.split_generator import SplitGenerator
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd.trainer import Trainer
    SubgraphSampler().run("job", SNC, None, uri_base=workdir)
    for with_split_files in (False, True):
        if with_split_files:
            SplitGenerator().run("job", SNC, None, uri_base=workdir)
        res = {}
        for route in ("tfrecord", "hbm"):
            doc = yaml.safe_load(open(os.path.join(workdir, SNC)))
            doc["trainerConfig"]["trainerArgs"]["data_route"] = route
            doc["sharedConfig"]["trainedModelMetadata"] = {"trainedModelUri": f"out/snc/model_{route}/model.pt",
                                                           "evalMetricsUri": f"out/snc/model_{route}/eval.json"}
            uri = SNC.replace(".yaml", f"_train_{route}.yaml")
            yaml.safe_dump(doc, open(os.path.join(workdir, uri), "w"))
            torch.manual_seed(5)
            tr = Trainer()
            with pytest.warns(RuntimeWarning) if not with_split_files else _nullcontext():
                metrics = tr.run("job", uri, None, uri_base=workdir)
            assert tr.training_process.route == route
            cfg = GbmlConfigPbWrapper.from_uri(uri, uri_base=workdir)
            res[route] = (torch.load(cfg.trained_model_uri, map_location="cpu"),
                          [h["loss"] for h in tr.training_process.trainer.history], metrics.metrics["acc"].value)
        (sd_t, loss_t, acc_t), (sd_h, loss_h, acc_h) = res["tfrecord"], res["hbm"]
        np.testing.assert_allclose(loss_h, loss_t, rtol=1e-4, atol=1e-5)
        assert acc_t == acc_h
        for k in sd_t:
            np.testing.assert_allclose(sd_h[k].numpy(), sd_t[k].numpy(), rtol=1e-3, atol=1e-4, err_msg=k)


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


@pytest.mark.gpu
def test_sample_with_replacement_takes_the_staged_forward_on_the_hbm_route(golden_dir, tmp_path):
    """experimentalFlags.sample_with_replacement = true (SGSPureSparkV1Task.scala:42-50,355-364): the one-call plan needs
    duplicate-free trees, so the in-HBM route runs such a job through the staged sample -> union -> forward instead of
    raising, and writes the rows the TFRecord route writes (same seed, same draws)"""
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd.trainer import Trainer
    base = str(tmp_path)
    shutil.copytree(os.path.join(golden_dir, "configs"), os.path.join(base, "configs"))
    shutil.copytree(os.path.join(golden_dir, "ref_assets"), os.path.join(base, "ref_assets"))
    doc = yaml.safe_load(open(os.path.join(base, SNC)))
    doc["datasetConfig"]["subgraphSamplerConfig"].setdefault("experimentalFlags", {})["sample_with_replacement"] = "true"
    yaml.safe_dump(doc, open(os.path.join(base, SNC), "w"))
    SubgraphSampler().run("job", SNC, None, uri_base=base)
    torch.manual_seed(1)
    Trainer().run("job", SNC, None, uri_base=base)
    a, b = Inferencer(), Inferencer()
    out_t = a.run("job", _variant(base, SNC, "tf"), None, uri_base=base, route="tfrecord")
    out_h = b.run("job", _variant(base, SNC, "auto"), None, uri_base=base)
    assert b.route == "hbm" and a.rows_written == b.rows_written == 16
    rt, rh = _rows(out_t["embeddings"]), _rows(out_h["embeddings"])
    assert [r["node_id"] for r in rt] == [r["node_id"] for r in rh]
    np.testing.assert_allclose(np.array([r["emb"] for r in rh], np.float32), np.array([r["emb"] for r in rt], np.float32),
                               rtol=1e-5, atol=1e-5)


# (batches that outgrow the one-call plan's workspace: tests/test_gpu_overflow.py)


@pytest.mark.gpu
def test_graph_data_batches_carry_the_union_graphs_csr():
    """ResidentGraph.graph_data packs the batch union graph's CSR into the GraphData it hands out instead of sorting the
    edge list again: the same rowptr / col / edge order as GraphData._build_csr gives for that edge_index, nodes and
    feature rows those of the oracle's collate"""
    import oracle
    from gigl_amd.engine import HipEngine
    from gigl_amd.hbm import ResidentGraph
    from gigl_amd.nn import GraphData
    from helpers import rmat_edges
    n, d = 1 << 11, 12
    s, t = rmat_edges(11, 40000, seed=9)
    rowptr, col = oracle.build_csc(n, s, t, is_directed=False)
    x = np.random.default_rng(2).standard_normal((n, d)).astype(np.float32)
    eng = HipEngine(0)
    try:
        eng.load_csc(rowptr, col)
        eng.load_features(x)
        res = ResidentGraph.from_engine(eng, np.arange(n), [7, 5])
        roots = np.random.default_rng(4).integers(0, n, size=200).astype(np.uint32)
        g, ri = res.graph_data(torch.from_numpy(roots.view(np.int32)).to(eng.device))
        ref = GraphData(x=g.x, edge_index=g.edge_index)
        ref._build_csr()
        assert torch.equal(g.rowptr, ref.rowptr) and torch.equal(g.col, ref.col) and torch.equal(g.n_dev, ref.n_dev)
        nbr, _ = oracle.sample_khop(rowptr, col, roots, [7, 5], canonical=True)
        u = oracle.union_build(roots, [7, 5], nbr)
        assert g.num_nodes == u["nodes"].size and g.num_edges == int(u["rowptr"][-1])
        assert np.array_equal(g.x.cpu().numpy(), x[u["nodes"].astype(np.int64)])
        assert np.array_equal(ri.cpu().numpy(), u["root_local"])
        assert np.array_equal(g.col.cpu().numpy()[: g.num_edges], u["col"][: g.num_edges])
    finally:
        eng.close()
