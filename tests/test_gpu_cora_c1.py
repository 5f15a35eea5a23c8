"""BASELINE.json configs[0] at its own size — SURVEY.md 8(d) C1: Cora-shaped graph (2,708 nodes, 5,278 undirected
random edges -> 10,556 directed), D = 1,433 bag-of-words rows (Bernoulli(0.0127), L1-normalised), 7 classes, fanout
[10,5], GraphSAGE 1433 -> 16 -> 7 (the node-classification spec's defaults,
/root/reference/python/gigl/src/common/modeling_task_specs/node_classification_modeling_task_spec.py:51-57: Adam lr 0.01,
weight decay 5e-4, main_sample_batch_size 16), B = 16 and B = 512 — exactly what `bench.py --workload cora` builds
(bench.build_workload / bench.cora_c1), run through
  * the one-call plan,
  * Inferencer.run on the in-HBM route and on the TFRecord route,
  * Trainer.run on the in-HBM route and on the TFRecord route,
each against the CPU oracle: sample (SamplingStrategy.scala:16-82 restated) -> collate (abstract_graph_builder.py
restated) -> fp32 forward / autograd (PyG SAGEConv formulas restated), at 1e-5 (forward) / 1e-4 (trained weights)."""
import argparse
import json
import os
import sys

import numpy as np
import pytest
import torch
import yaml

import oracle
from oracle import gnn_ref

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

FAN = [10, 5]


@pytest.fixture(scope="module")
def c1():
    import bench
    n, src, dst, x, labels = bench.cora_c1(1)
    rowptr, col = oracle.build_csc(n, src.astype(np.uint32), dst.astype(np.uint32), is_directed=False)
    assert n == 2708 and col.size == 10556 and x.shape == (2708, 1433)
    return n, src, dst, x, labels, rowptr, col


def _oracle_rows(rowptr, col, x, roots, sd):
    nbr, _ = oracle.sample_khop(rowptr, col, roots, FAN, canonical=True)
    u = oracle.union_build(roots, FAN, nbr)
    ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
    o = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"].astype(np.int64)]), ei, sd, 2)
    return nbr, o[u["root_local"]].numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("b", [16, 512])
def test_cora_c1_through_the_one_call_plan(c1, b):
    """bench.build_workload("cora") -> GraphSAGE(1433, 16, 7).make_plan: sampled trees bit-equal to the oracle's, root
    rows within 1e-5 of the fp32 CPU forward over the oracle-collated batch; several batches of the seed-42 root order"""
    import bench
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE
    n, src, dst, x, labels, rowptr, col = c1
    eng = HipEngine(0)
    try:
        args = argparse.Namespace(workload="cora")
        assert bench.build_workload(eng, args) == (2708, 1433)
        assert args._workload[2:4] == (16, 7)
        torch.manual_seed(0)
        model = GraphSAGE(1433, 16, 7, num_layers=2).to(eng.device)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        plan = model.make_plan(eng, b, FAN)
        order = np.random.RandomState(42).permutation(n).astype(np.uint32)
        for it in range(3):
            roots = order[it * b:(it + 1) * b]
            out = plan.run(torch.from_numpy(roots.view(np.int32)).to(eng.device)).cpu().numpy()
            nbr, want = _oracle_rows(rowptr, col, x, roots, sd)
            hb = plan.last_batch_to_host()
            for k in range(2):
                assert np.array_equal(hb["nbr"][k], nbr[k])
            np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-5)
        plan.close()
    finally:
        eng.close()


def _write_c1_job(base, c1, max_train=64):
    """C1 as the Data Preprocessor's tables + a frozen task config with the spec's defaults"""
    from gigl_amd import wire
    n, src, dst, x, labels, _, _ = c1
    os.makedirs(os.path.join(base, "tables/nodes"), exist_ok=True)
    os.makedirs(os.path.join(base, "tables/edges"), exist_ok=True)
    wire.write_tfrecords(os.path.join(base, "tables/nodes/data.tfrecord"), [
        wire.encode_tf_example({"node_id": np.array([i], np.int64), "feat": x[i],
                                "node_label": np.array([labels[i]], np.int64)}) for i in range(n)])
    wire.write_tfrecords(os.path.join(base, "tables/edges/data.tfrecord"), [
        wire.encode_tf_example({"src": np.array([s], np.int64), "dst": np.array([t], np.int64)})
        for s, t in zip(src.tolist(), dst.tolist())])
    os.makedirs(os.path.join(base, "configs"), exist_ok=True)
    yaml.safe_dump({
        "condensedEdgeTypeToPreprocessedMetadata": {"0": {"dstNodeIdKey": "dst", "srcNodeIdKey": "src", "mainEdgeInfo": {
            "tfrecordUriPrefix": "tables/edges", "featureDim": 0}}},
        "condensedNodeTypeToPreprocessedMetadata": {"0": {"featureDim": 1433, "featureKeys": ["feat"], "labelKeys": [
            "node_label"], "nodeIdKey": "node_id", "tfrecordUriPrefix": "tables/nodes"}}},
        open(os.path.join(base, "configs/pm.yaml"), "w"))
    spec = "gigl_amd.task_specs.HipGraphSageNodeClassificationSpec"
    edge_type = {"srcNodeType": "paper", "relation": "cites", "dstNodeType": "paper"}
    for route in ("hbm", "tfrecord"):
        yaml.safe_dump({
            "graphMetadata": {"edgeTypes": [edge_type], "nodeTypes": ["paper"]},
            "taskMetadata": {"nodeBasedTaskMetadata": {"supervisionNodeTypes": ["paper"]}},
            "datasetConfig": {"subgraphSamplerConfig": {
                "numHops": 2, "numNeighborsToSample": 10, "numMaxTrainingSamplesToOutput": max_train,
                "experimentalFlags": {"permutation_strategy": "deterministic"},
                "subgraphSamplingStrategy": {"messagePassingPaths": {"paths": [{"rootNodeType": "paper", "samplingOps": [
                    {"opName": f"hop{k}", "edgeType": edge_type, "randomUniform": {"numNodesToSample": f},
                     "inputOpNames": ([f"hop{k - 1}"] if k else [])} for k, f in enumerate(FAN)]}]}}}},
            "sharedConfig": {
                "isGraphDirected": False,
                "flattenedGraphMetadata": {"supervisedNodeClassificationOutput": {
                    "labeledTfrecordUriPrefix": "out/labeled/samples/", "unlabeledTfrecordUriPrefix": "out/unlabeled/samples/"}},
                "preprocessedMetadataUri": "configs/pm.yaml",
                "trainedModelMetadata": {"trainedModelUri": f"out/model_{route}/model.pt",
                                         "evalMetricsUri": f"out/model_{route}/eval.json"},
                "inferenceMetadata": {"nodeTypeToInferencerOutputInfoMap": {"paper": {
                    "embeddingsPath": f"out/inference_{route}/embeddings.jsonl",
                    "predictionsPath": f"out/inference_{route}/predictions.jsonl"}}}},
            # (every other argument is the spec's default: hid 16, out 7, lr 0.01, wd 5e-4, batch 16)
            "trainerConfig": {"trainerClsPath": spec, "trainerArgs": {"num_epochs": "2", "data_route": route}},
            "inferencerConfig": {"inferencerClsPath": spec, "inferencerArgs": {"data_route": route},
                                 "inferenceBatchSize": 512}},
            open(os.path.join(base, f"configs/job_{route}.yaml"), "w"))


def _cpu_training(c1, roots_all, labels_all, b, epochs, seed):
    """NodeClassificationModelingTaskSpec._train restated on the CPU: batches of b consecutive labeled roots, oracle
    sample -> collate, fp32 forward with autograd, cross-entropy on the roots, Adam(lr 0.01, weight_decay 5e-4)"""
    from gigl_amd.models import GraphSAGE
    n, src, dst, x, labels, rowptr, col = c1
    torch.manual_seed(seed)
    init = GraphSAGE(1433, 16, 7, num_layers=2).state_dict()
    params = {k: v.detach().clone().requires_grad_(True) for k, v in init.items()}
    opt = torch.optim.Adam(list(params.values()), lr=0.01, weight_decay=5e-4)
    last = []
    for _ in range(epochs):
        for lo in range(0, roots_all.size, b):
            roots = roots_all[lo:lo + b].astype(np.uint32)
            nbr, _ = oracle.sample_khop(rowptr, col, roots, FAN, canonical=True)
            u = oracle.union_build(roots, FAN, nbr)
            ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
            out = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"].astype(np.int64)]), ei, params, 2)
            loss = torch.nn.functional.cross_entropy(out[torch.from_numpy(u["root_local"].astype(np.int64))],
                                                     torch.from_numpy(labels_all[lo:lo + b]))
            opt.zero_grad()
            loss.backward()
            opt.step()
        last.append(float(loss))
    return {k: v.detach() for k, v in params.items()}, last


@pytest.mark.gpu
def test_cora_c1_through_the_entry_points(c1, tmp_path_factory):
    from gigl_amd.config import GbmlConfigPbWrapper
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd.trainer import Trainer
    n, src, dst, x, labels, rowptr, col = c1
    base = str(tmp_path_factory.mktemp("cora_c1"))
    _write_c1_job(base, c1)
    SubgraphSampler().run("job", "configs/job_tfrecord.yaml", None, uri_base=base)
    # ---- Trainer: both routes against the CPU restatement (64 labeled roots = 4 steps per epoch, 2 epochs)
    deg = np.diff(rowptr)
    train_roots = np.flatnonzero(deg > 0)[:64].astype(np.int64)
    sd_cpu, loss_cpu = _cpu_training(c1, train_roots, labels[train_roots], 16, 2, seed=5)
    trained = {}
    for route in ("hbm", "tfrecord"):
        torch.manual_seed(5)
        tr = Trainer()
        with pytest.warns(RuntimeWarning):  # (no split-generator output: the root-id split, whole for < 100 samples)
            tr.run("job", f"configs/job_{route}.yaml", None, uri_base=base)
        assert tr.training_process.route == route
        cfg = GbmlConfigPbWrapper.from_uri(f"configs/job_{route}.yaml", uri_base=base)
        sd = torch.load(cfg.trained_model_uri, map_location="cpu")
        hist = [h["loss"] for h in tr.training_process.trainer.history]
        np.testing.assert_allclose(hist, loss_cpu, rtol=1e-4, atol=1e-5, err_msg=route)
        for k in sd_cpu:
            np.testing.assert_allclose(sd[k].numpy(), sd_cpu[k].numpy(), rtol=1e-3, atol=1e-4, err_msg=f"{route}: {k}")
        trained[route] = sd
    # ---- Inferencer: both routes, rows against the oracle over the batches the route walks (512 roots per batch)
    rows = {}
    for route in ("hbm", "tfrecord"):
        inf = Inferencer()
        out = inf.run("job", f"configs/job_{route}.yaml", None, uri_base=base)
        assert inf.route == route and inf.rows_written == n
        rows[route] = [json.loads(l) for l in open(out["embeddings"])]
    ids = [r["node_id"] for r in rows["hbm"]]
    assert ids == [r["node_id"] for r in rows["tfrecord"]] and sorted(ids) == list(range(n))
    for route in ("hbm", "tfrecord"):
        emb = np.array([r["emb"] for r in rows[route]], np.float32)
        for lo in range(0, n, 512):
            roots = np.array(ids[lo:lo + 512], dtype=np.uint32)
            _, want = _oracle_rows(rowptr, col, x, roots, trained[route])
            np.testing.assert_allclose(emb[lo:lo + 512], want, rtol=1e-5, atol=1e-5, err_msg=f"{route} batch at {lo}")
