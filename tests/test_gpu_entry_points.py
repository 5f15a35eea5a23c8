"""Drop-in entry points end to end on the reference's own fixture graph: SubgraphSampler.run -> Trainer.run ->
Inferencer.run, plus autograd parity of the HIP SAGE layer against the fp32 CPU restatement."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

from conftest import seed_trainer

import oracle
from gigl_amd import wire
from gigl_amd.config import GbmlConfigPbWrapper, tfrecord_files
from gigl_amd.sampler_service import build_rooted_node_neighborhood, tree_to_edge_lists
from helpers import check_rnn_validity, load_fixture_graph, rmat_edges

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir(golden_dir, tmp_path_factory):
    """a scratch uri_base holding the configs + reference input assets (outputs land next to them)"""
    base = tmp_path_factory.mktemp("gigl_e2e")
    shutil.copytree(os.path.join(golden_dir, "configs"), base / "configs")
    shutil.copytree(os.path.join(golden_dir, "ref_assets"), base / "ref_assets")
    return str(base)


def test_subgraph_sampler_node_classification(workdir, golden_dir):
    from gigl_amd.subgraph_sampler import SubgraphSampler
    files = SubgraphSampler().run("job", "configs/snc_frozen_gbml_config.yaml", None, uri_base=workdir)
    cfg = GbmlConfigPbWrapper.from_uri("configs/snc_frozen_gbml_config.yaml", uri_base=workdir)
    unl = [wire.RootedNodeNeighborhood.FromString(r) for f in tfrecord_files(cfg.unlabeled_tfrecord_uri_prefix)
           for r in wire.read_tfrecords(f)]
    lab = [wire.SupervisedNodeClassificationSample.FromString(r) for f in tfrecord_files(cfg.labeled_tfrecord_uri_prefix)
           for r in wire.read_tfrecords(f)]
    # same record counts as the reference's real sampler output for this graph: 16 RNN, 14 labeled
    assert len(unl) == 16 and len(lab) == 14 and files["unlabeled"] and files["labeled"]
    n, src, dst, feats = load_fixture_graph(golden_dir)
    rowptr, col = oracle.build_csc(n, src, dst, is_directed=False)
    roots = np.arange(n, dtype=np.uint32)
    nbr_o, _ = oracle.sample_khop(rowptr, col, roots, [3, 3], canonical=True)
    want = [build_rooted_node_neighborhood(r, s, d, feats).SerializeToString()
            for r, (s, d) in zip(roots.tolist(), tree_to_edge_lists(roots, [3, 3], nbr_o))]
    assert [u.SerializeToString() for u in unl] == want  # byte-identical to the oracle-derived records
    for u in unl:
        check_rnn_validity(u.root_node.node_id, [(e.src_node_id, e.dst_node_id) for e in u.neighborhood.edges],
                           [x.node_id for x in u.neighborhood.nodes], rowptr, col, fanout=3)
    assert sorted(s.root_node.node_id for s in lab) == [i for i in range(16) if i not in (14, 15)]
    assert all(s.root_node_labels and s.root_node_labels[0].label_type == "node_label" for s in lab)


def test_subgraph_sampler_link_prediction(workdir):
    from gigl_amd.subgraph_sampler import SubgraphSampler
    SubgraphSampler().run("job", "configs/nablp_frozen_gbml_config.yaml", None, uri_base=workdir)
    cfg = GbmlConfigPbWrapper.from_uri("configs/nablp_frozen_gbml_config.yaml", uri_base=workdir)
    rn = [wire.RootedNodeNeighborhood.FromString(r) for p in cfg.random_negative_tfrecord_uri_prefixes.values()
          for f in tfrecord_files(p) for r in wire.read_tfrecords(f)]
    smp = [wire.NodeAnchorBasedLinkPredictionSample.FromString(r) for f in tfrecord_files(cfg.nablp_tfrecord_uri_prefix)
           for r in wire.read_tfrecords(f)]
    assert len(rn) == 27  # every node of the toy graph, isolated ones included
    assert smp and all(1 <= len(s.pos_edges) <= 2 for s in smp)
    by_root = {r.root_node.node_id: r for r in rn}
    for s in smp:
        ids = {x.node_id for x in s.neighborhood.nodes}
        assert s.root_node.node_id in ids
        for e in s.pos_edges:  # supervision-edge endpoints are in the neighbourhood (subgraph_sampler_test.py:529-684)
            assert e.src_node_id == s.root_node.node_id and e.dst_node_id in ids
        for e in s.neighborhood.edges:
            assert e.src_node_id in ids and e.dst_node_id in ids
        # neighbourhood = root's rooted sample merged with each positive's rooted sample
        want = set((e.src_node_id, e.dst_node_id) for e in by_root[s.root_node.node_id].neighborhood.edges)
        for pe in s.pos_edges:
            want |= set((e.src_node_id, e.dst_node_id) for e in by_root[pe.dst_node_id].neighborhood.edges)
        assert set((e.src_node_id, e.dst_node_id) for e in s.neighborhood.edges) == want
        assert not s.hard_neg_edges and not s.neg_edges


def test_sage_layer_autograd_matches_cpu_reference():
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE
    from gigl_amd.nn import GraphData
    from oracle import gnn_ref
    eng = HipEngine(0)
    torch.manual_seed(0)
    n, e, d = 500, 3000, 24
    ei = torch.randint(0, n, (2, e))
    x = torch.randn(n, d)
    model = GraphSAGE(d, 32, 5, num_layers=2)
    ref_params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    ref_out = gnn_ref.graphsage_forward(x, ei, ref_params, 2)
    target = torch.randint(0, 5, (n,))
    ref_loss = torch.nn.functional.cross_entropy(ref_out, target)
    ref_loss.backward()
    dev = eng.device
    model = model.to(dev)
    model.engine = eng
    g = GraphData(x=x, edge_index=ei).to(dev)
    out = model(g)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out.detach().numpy(), rtol=1e-5, atol=1e-5)
    loss = torch.nn.functional.cross_entropy(out, target.to(dev))
    loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-5
    for name, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref_params[name].grad.numpy(), rtol=1e-4, atol=1e-6,
                                   err_msg=name)
    eng.close()


def test_trainer_then_inferencer(workdir, golden_dir):
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd.trainer import Trainer
    from oracle import gnn_ref
    cfg_uri = "configs/snc_frozen_gbml_config.yaml"
    SubgraphSampler().run("job", cfg_uri, None, uri_base=workdir)
    seed_trainer()  # (ONE shared seed for every trainer test: tests/conftest.py; the trainer seeds nothing itself)
    tr = Trainer()
    metrics = tr.run("job", cfg_uri, None, uri_base=workdir)
    assert "acc" in metrics.metrics and 0.0 <= metrics.metrics["acc"].value <= 1.0
    spec = tr.training_process.trainer
    hist = spec.history
    assert len(hist) == 3 and all(np.isfinite(h["loss"]) for h in hist)
    assert hist[-1]["loss"] < hist[0]["loss"]  # it learns the 14-sample fixture
    cfg = GbmlConfigPbWrapper.from_uri(cfg_uri, uri_base=workdir)
    sd = torch.load(cfg.trained_model_uri, map_location="cpu")
    assert set(sd) == {"conv_layers.0.lin_l.weight", "conv_layers.0.lin_l.bias", "conv_layers.0.lin_r.weight",
                       "conv_layers.1.lin_l.weight", "conv_layers.1.lin_l.bias", "conv_layers.1.lin_r.weight"}
    assert json.load(open(cfg.eval_metrics_uri))["metrics"][0]["name"] == "acc"
    inf = Inferencer()
    out = inf.run("job", cfg_uri, None, uri_base=workdir)
    rows = [json.loads(l) for l in open(out["embeddings"])]
    preds = [json.loads(l) for l in open(out["predictions"])]
    assert inf.rows_written == 16 and len(rows) == 16 and len(preds) == 16
    # per-root embedding == fp32 CPU forward of the trained weights over the same batches (inferenceBatchSize 8)
    from gigl_amd.batches import RootedNodeNeighborhoodBatch, iterate_tfrecord_batches
    want = {}
    for raw in iterate_tfrecord_batches(tfrecord_files(cfg.unlabeled_tfrecord_uri_prefix), 8):
        b = RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(raw)
        o = gnn_ref.graphsage_forward(b.graph.x, b.graph.edge_index, sd, 2)
        for r, i in zip(b.root_nodes, b.condensed_node_type_to_root_node_indices_map[0].tolist()):
            want[r.id] = o[i].numpy()
    for row, p in zip(rows, preds):
        np.testing.assert_allclose(np.array(row["emb"], np.float32), want[row["node_id"]], rtol=1e-5, atol=1e-5)
        assert p["node_id"] == row["node_id"] and p["pred"] == int(np.argmax(want[row["node_id"]]))


def test_inferencer_avro_embedding_shards(workdir):
    """an embeddingsPath that names a directory gets the reference exporter's Avro shards (device-encoded); the decoded
    records equal the line-per-root JSON output of the same run configuration"""
    import yaml
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd.trainer import Trainer
    from oracle import avro
    cfg_uri = "configs/snc_frozen_gbml_config.yaml"
    SubgraphSampler().run("job", cfg_uri, None, uri_base=workdir)
    Trainer().run("job", cfg_uri, None, uri_base=workdir)
    out_json = Inferencer().run("job", cfg_uri, None, uri_base=workdir)
    rows = [json.loads(l) for l in open(out_json["embeddings"])]
    doc = yaml.safe_load(open(os.path.join(workdir, cfg_uri)))
    doc["sharedConfig"]["inferenceMetadata"]["nodeTypeToInferencerOutputInfoMap"]["user"]["embeddingsPath"] = \
        "out/snc/inference/embeddings_avro/"
    avro_cfg = "configs/snc_avro_gbml_config.yaml"
    yaml.safe_dump(doc, open(os.path.join(workdir, avro_cfg), "w"))
    out = Inferencer().run("job", avro_cfg, None, uri_base=workdir)
    shards = sorted(os.listdir(out["embeddings"]))
    assert shards == ["shard_00000000.avro"]
    schema, recs = avro.read_embedding_file(open(os.path.join(out["embeddings"], shards[0]), "rb").read())
    assert [f["name"] for f in schema["fields"]] == ["node_id", "node_type", "emb"]
    assert len(recs) == 16 and [r["node_id"] for r in recs] == [r["node_id"] for r in rows]
    for a, b in zip(recs, rows):
        assert a["node_type"] == "user"
        np.testing.assert_array_equal(np.array(a["emb"], np.float32), np.array(b["emb"], np.float32))


def test_subgraph_sampler_hydrates_edge_features(workdir, golden_dir):
    """mainEdgeInfo.featureKeys -> `_edge_features` (SGSPureSparkV1Task.scala:172-193) -> Edge.feature_values of every
    neighbourhood edge (hydrateEdges :549-593): the fixture's edge table re-written with two feature columns (a scalar
    and a 2-vector, symmetric in the endpoints because the graph is bidirectionalised)"""
    import yaml
    from gigl_amd.subgraph_sampler import SubgraphSampler
    src_dir = os.path.join(workdir, "ref_assets/subgraph_sampler/supervised_node_classification/edge_data")
    rows = [wire.decode_tf_example(r) for f in tfrecord_files(src_dir + "/") for r in wire.read_tfrecords(f)]

    def feats(a, b):
        lo, hi = min(a, b), max(a, b)
        return np.float32(lo * 10 + hi), np.array([lo - hi, 0.5 * hi], np.float32)
    ef_dir = os.path.join(workdir, "edge_data_with_features")
    os.makedirs(ef_dir, exist_ok=True)
    out = []
    for r in rows:
        a, b = int(np.asarray(r["src"]).ravel()[0]), int(np.asarray(r["dst"]).ravel()[0])
        w, v = feats(a, b)
        out.append(wire.encode_tf_example({"src": np.array([a], np.int64), "dst": np.array([b], np.int64),
                                           "w": np.array([w], np.float32), "v": v}))
    wire.write_tfrecords(os.path.join(ef_dir, "data.tfrecord"), out)
    pm = yaml.safe_load(open(os.path.join(workdir, "configs/snc_preprocessed_metadata.yaml")))
    em = pm["condensedEdgeTypeToPreprocessedMetadata"]["0"]
    em["mainEdgeInfo"].update(tfrecordUriPrefix="edge_data_with_features", featureKeys=["w", "v"], featureDim=3)
    yaml.safe_dump(pm, open(os.path.join(workdir, "configs/snc_ef_preprocessed_metadata.yaml"), "w"))
    doc = yaml.safe_load(open(os.path.join(workdir, "configs/snc_frozen_gbml_config.yaml")))
    doc["sharedConfig"]["preprocessedMetadataUri"] = "configs/snc_ef_preprocessed_metadata.yaml"
    flat = doc["sharedConfig"]["flattenedGraphMetadata"]["supervisedNodeClassificationOutput"]
    flat["labeledTfrecordUriPrefix"] = "out/snc_ef/labeled/samples/"
    flat["unlabeledTfrecordUriPrefix"] = "out/snc_ef/unlabeled/samples/"
    yaml.safe_dump(doc, open(os.path.join(workdir, "configs/snc_ef_gbml_config.yaml"), "w"))
    SubgraphSampler().run("job", "configs/snc_ef_gbml_config.yaml", None, uri_base=workdir)
    cfg = GbmlConfigPbWrapper.from_uri("configs/snc_ef_gbml_config.yaml", uri_base=workdir)
    plain = GbmlConfigPbWrapper.from_uri("configs/snc_frozen_gbml_config.yaml", uri_base=workdir)
    SubgraphSampler().run("job", "configs/snc_frozen_gbml_config.yaml", None, uri_base=workdir)
    got = [wire.RootedNodeNeighborhood.FromString(r) for f in tfrecord_files(cfg.unlabeled_tfrecord_uri_prefix)
           for r in wire.read_tfrecords(f)]
    base = [wire.RootedNodeNeighborhood.FromString(r) for f in tfrecord_files(plain.unlabeled_tfrecord_uri_prefix)
            for r in wire.read_tfrecords(f)]
    assert len(got) == len(base) == 16
    n_edges = 0
    for g, b in zip(got, base):  # same sample, plus the features
        assert g.root_node == b.root_node and g.neighborhood.nodes == b.neighborhood.nodes
        assert [(e.src_node_id, e.dst_node_id) for e in g.neighborhood.edges] == \
               [(e.src_node_id, e.dst_node_id) for e in b.neighborhood.edges]
        for e in g.neighborhood.edges:
            w, v = feats(e.src_node_id, e.dst_node_id)
            np.testing.assert_array_equal(e.feature_values, np.concatenate([[w], v]).astype(np.float32))
            n_edges += 1
    assert n_edges > 20
    lab = [wire.SupervisedNodeClassificationSample.FromString(r) for f in tfrecord_files(cfg.labeled_tfrecord_uri_prefix)
           for r in wire.read_tfrecords(f)]
    assert len(lab) == 14 and all(e.feature_values.size == 3 for s_ in lab for e in s_.neighborhood.edges)


def test_node_classification_with_gat_encoder(workdir):
    """gnn_model_class_path swaps the encoder of the node-classification plugin: a 2-head GAT trains through the HIP
    backward, is saved with PyG GATConv's parameter names, and the inferencer's predictions equal the fp32 restatement"""
    import yaml
    from gigl_amd.batches import RootedNodeNeighborhoodBatch, iterate_tfrecord_batches
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd.trainer import Trainer
    from oracle import gnn_ref
    base = "configs/snc_frozen_gbml_config.yaml"
    SubgraphSampler().run("job", base, None, uri_base=workdir)
    doc = yaml.safe_load(open(os.path.join(workdir, base)))
    for sect, key in (("trainerConfig", "trainerArgs"), ("inferencerConfig", "inferencerArgs")):
        doc[sect][key].update(gnn_model_class_path="gigl_amd.models_attn.GAT", num_heads="2", hid_dim="8", out_dim="4")
    doc["sharedConfig"]["trainedModelMetadata"].update(trainedModelUri="out/snc_gat/model.pt",
                                                        evalMetricsUri="out/snc_gat/eval_metrics.json")
    info = doc["sharedConfig"]["inferenceMetadata"]["nodeTypeToInferencerOutputInfoMap"]["user"]
    info.update(embeddingsPath="out/snc_gat/embeddings.jsonl", predictionsPath="out/snc_gat/predictions.jsonl")
    cfg_uri = "configs/snc_gat_gbml_config.yaml"
    yaml.safe_dump(doc, open(os.path.join(workdir, cfg_uri), "w"))
    seed_trainer()  # (ONE shared seed for every trainer test: tests/conftest.py; the trainer seeds nothing itself)
    tr = Trainer()
    tr.run("job", cfg_uri, None, uri_base=workdir)
    hist = tr.training_process.trainer.history
    assert all(np.isfinite(h["loss"]) for h in hist) and hist[-1]["loss"] < hist[0]["loss"]
    cfg = GbmlConfigPbWrapper.from_uri(cfg_uri, uri_base=workdir)
    sd = torch.load(cfg.trained_model_uri, map_location="cpu")
    assert {"conv_layers.0.lin.weight", "conv_layers.0.att_src", "conv_layers.0.att_dst", "conv_layers.0.bias",
            "conv_layers.1.lin.weight"} <= set(sd) and sd["conv_layers.0.lin.weight"].shape[0] == 16
    out = Inferencer().run("job", cfg_uri, None, uri_base=workdir)
    rows = [json.loads(l) for l in open(out["embeddings"])]
    preds = [json.loads(l) for l in open(out["predictions"])]
    want = {}
    for raw in iterate_tfrecord_batches(tfrecord_files(cfg.unlabeled_tfrecord_uri_prefix), 8):
        b = RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(raw)
        h = b.graph.x
        for l in range(2):
            p = f"conv_layers.{l}."
            h = gnn_ref.gat_conv(h, b.graph.edge_index, sd[p + "lin.weight"], sd[p + "att_src"], sd[p + "att_dst"],
                                 sd[p + "bias"], 2 if l == 0 else 1)
            if l == 0:
                h = torch.relu(h)
        for r, i in zip(b.root_nodes, b.condensed_node_type_to_root_node_indices_map[0].tolist()):
            want[r.id] = h[i].numpy()
    assert len(rows) == 16
    for row, p in zip(rows, preds):
        np.testing.assert_allclose(np.array(row["emb"], np.float32), want[row["node_id"]], rtol=1e-5, atol=1e-5)
        assert p["pred"] == int(np.argmax(want[row["node_id"]]))


def test_node_classification_with_the_stock_two_layer_gcn(workdir):
    """the reference's default node-classification model (TwoLayerGCN) behind the same plugin: trains through the HIP
    kernels (GCN aggregation forward + its transpose backward, MFMA projections) and serves the inferencer"""
    import yaml
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd.trainer import Trainer
    base = "configs/snc_frozen_gbml_config.yaml"
    SubgraphSampler().run("job", base, None, uri_base=workdir)
    doc = yaml.safe_load(open(os.path.join(workdir, base)))
    for sect, key in (("trainerConfig", "trainerArgs"), ("inferencerConfig", "inferencerArgs")):
        doc[sect][key].update(gnn_model_class_path="gigl_amd.models_attn.TwoLayerGCN", hid_dim="8", out_dim="3")
    doc["trainerConfig"]["trainerArgs"]["num_epochs"] = "6"
    doc["sharedConfig"]["trainedModelMetadata"].update(trainedModelUri="out/snc_gcn/model.pt",
                                                        evalMetricsUri="out/snc_gcn/eval_metrics.json")
    info = doc["sharedConfig"]["inferenceMetadata"]["nodeTypeToInferencerOutputInfoMap"]["user"]
    info.update(embeddingsPath="out/snc_gcn/embeddings.jsonl", predictionsPath="out/snc_gcn/predictions.jsonl")
    cfg_uri = "configs/snc_gcn_gbml_config.yaml"
    yaml.safe_dump(doc, open(os.path.join(workdir, cfg_uri), "w"))
    seed_trainer()  # (ONE shared seed for every trainer test: tests/conftest.py; the trainer seeds nothing itself)
    tr = Trainer()
    metrics = tr.run("job", cfg_uri, None, uri_base=workdir)
    hist = tr.training_process.trainer.history
    assert len(hist) == 6 and all(np.isfinite(h["loss"]) for h in hist) and min(h["loss"] for h in hist[1:]) < hist[0]["loss"]
    assert 0.0 <= metrics.metrics["acc"].value <= 1.0
    cfg = GbmlConfigPbWrapper.from_uri(cfg_uri, uri_base=workdir)
    sd = torch.load(cfg.trained_model_uri, map_location="cpu")
    assert set(sd) == {"conv1.lin.weight", "conv1.bias", "conv2.lin.weight", "conv2.bias"}  # PyG GCNConv's names
    inf = Inferencer()
    inf.run("job", cfg_uri, None, uri_base=workdir)
    assert inf.rows_written == 16


def test_non_deterministic_strategy_and_with_replacement_flag(workdir):
    """experimental_flags.permutation_strategy other than "deterministic" = the reference's F.shuffle: served with a
    fresh random seed (valid uniform samples, no parity); sample_with_replacement draws numNeighborsToSample
    neighbours per parent with replacement (valid edges, every non-isolated parent fully drawn)"""
    import yaml
    from gigl_amd.subgraph_sampler import SubgraphSampler
    doc = yaml.safe_load(open(os.path.join(workdir, "configs/snc_frozen_gbml_config.yaml")))
    flags = doc["datasetConfig"]["subgraphSamplerConfig"]["experimentalFlags"]
    flags["permutation_strategy"] = "non-deterministic"
    flat = doc["sharedConfig"]["flattenedGraphMetadata"]["supervisedNodeClassificationOutput"]
    flat["labeledTfrecordUriPrefix"] = "out/snc_nd/labeled/samples/"
    flat["unlabeledTfrecordUriPrefix"] = "out/snc_nd/unlabeled/samples/"
    yaml.safe_dump(doc, open(os.path.join(workdir, "configs/snc_nd_gbml_config.yaml"), "w"))
    sg = SubgraphSampler()
    sg.run("job", "configs/snc_nd_gbml_config.yaml", None, uri_base=workdir)
    assert sg.sampling_seed != 42 and 1 <= sg.sampling_seed < (1 << 20)
    cfg = GbmlConfigPbWrapper.from_uri("configs/snc_nd_gbml_config.yaml", uri_base=workdir)
    recs = [wire.RootedNodeNeighborhood.FromString(r) for f in tfrecord_files(cfg.unlabeled_tfrecord_uri_prefix)
            for r in wire.read_tfrecords(f)]
    assert len(recs) == 16
    # validity of the reference's output validator + fanout bound (3 per hop, 2 hops)
    edge_rows = [wire.decode_tf_example(r) for f in tfrecord_files(os.path.join(
        workdir, "ref_assets/subgraph_sampler/supervised_node_classification/edge_data/")) for r in wire.read_tfrecords(f)]
    und = {(int(np.ravel(r["src"])[0]), int(np.ravel(r["dst"])[0])) for r in edge_rows}
    und |= {(b, a) for a, b in und}
    for m in recs:
        ids = {x.node_id for x in m.neighborhood.nodes}
        assert m.root_node.node_id in ids and len(m.neighborhood.edges) <= 3 + 9
        for e in m.neighborhood.edges:
            assert (e.src_node_id, e.dst_node_id) in und and e.src_node_id in ids and e.dst_node_id in ids
    flags["sample_with_replacement"] = "true"
    yaml.safe_dump(doc, open(os.path.join(workdir, "configs/snc_wr_gbml_config.yaml"), "w"))
    flat["labeledTfrecordUriPrefix"] = "out/snc_wr/labeled/samples/"
    flat["unlabeledTfrecordUriPrefix"] = "out/snc_wr/unlabeled/samples/"
    yaml.safe_dump(doc, open(os.path.join(workdir, "configs/snc_wr_gbml_config.yaml"), "w"))
    SubgraphSampler().run("job", "configs/snc_wr_gbml_config.yaml", None, uri_base=workdir)
    cfg = GbmlConfigPbWrapper.from_uri("configs/snc_wr_gbml_config.yaml", uri_base=workdir)
    recs = [wire.RootedNodeNeighborhood.FromString(r) for f in tfrecord_files(cfg.unlabeled_tfrecord_uri_prefix)
            for r in wire.read_tfrecords(f)]
    assert len(recs) == 16
    deg = {}
    for a, b in und:
        deg[b] = deg.get(b, 0) + 1
    for m in recs:
        ids = {x.node_id for x in m.neighborhood.nodes}
        r = m.root_node.node_id
        hop1 = [e for e in m.neighborhood.edges if e.dst_node_id == r]
        for e in m.neighborhood.edges:  # every drawn edge exists, endpoints are in the neighbourhood
            assert (e.src_node_id, e.dst_node_id) in und and e.src_node_id in ids and e.dst_node_id in ids
        if deg.get(r, 0):  # sampleWithReplacementUDF: exactly numSamples draws (repeats kept as repeated edges)
            assert len(hop1) >= 3 and len(m.neighborhood.edges) >= 3 + 3


def test_sampler_split_generator_trainer_chain(workdir):
    """sampler -> split generator -> trainer: the trainer reads the train/val/test files the split generator wrote
    (datasetMetadata.supervisedNodeClassificationDataset), as the reference's pipeline does"""
    from gigl_amd.split_generator import NodeToDatasetSplitHashingAssigner, SplitGenerator
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd.trainer import Trainer
    cfg_uri = "configs/snc_frozen_gbml_config.yaml"
    SubgraphSampler().run("job", cfg_uri, None, uri_base=workdir)
    files = SplitGenerator().run("job", cfg_uri, None, uri_base=workdir)["main"]
    assigner = NodeToDatasetSplitHashingAssigner({"train_split": "0.4", "val_split": "0.3", "test_split": "0.3"})
    roots = {}
    for split, fs in files.items():
        recs = [wire.SupervisedNodeClassificationSample.FromString(r) for f in fs for r in wire.read_tfrecords(f)]
        roots[split] = sorted(s.root_node.node_id for s in recs)
        assert all(assigner.assign(s.root_node) == split for s in recs)
    assert sorted(sum(roots.values(), [])) == [i for i in range(16) if i not in (14, 15)]  # each labeled sample once
    assert all(roots[s] for s in ("train", "val", "test"))
    seed_trainer()  # (ONE shared seed for every trainer test: tests/conftest.py; the trainer seeds nothing itself)
    tr = Trainer()
    metrics = tr.run("job", cfg_uri, None, uri_base=workdir)
    hist = tr.training_process.trainer.history
    assert len(hist) == 3 and all(np.isfinite(h["loss"]) for h in hist)
    assert 0.0 <= metrics.metrics["acc"].value <= 1.0


def test_subgraph_sampler_caps_training_samples(workdir):
    """numMaxTrainingSamplesToOutput (downsampleNumberOfNodes): at most n labeled samples, all nodes keep their
    RootedNodeNeighborhood"""
    import yaml
    from gigl_amd.subgraph_sampler import SubgraphSampler
    doc = yaml.safe_load(open(os.path.join(workdir, "configs/snc_frozen_gbml_config.yaml")))
    doc["datasetConfig"]["subgraphSamplerConfig"]["numMaxTrainingSamplesToOutput"] = 5
    doc["sharedConfig"]["flattenedGraphMetadata"]["supervisedNodeClassificationOutput"] = {
        "labeledTfrecordUriPrefix": "out/snc_cap/labeled/samples/", "unlabeledTfrecordUriPrefix": "out/snc_cap/unlabeled/samples/"}
    with open(os.path.join(workdir, "configs/snc_cap.yaml"), "w") as fh:
        yaml.safe_dump(doc, fh)
    files = SubgraphSampler().run("job", "configs/snc_cap.yaml", None, uri_base=workdir, batch_size=4)
    lab = [wire.SupervisedNodeClassificationSample.FromString(r) for f in files["labeled"] for r in wire.read_tfrecords(f)]
    unl = [r for f in files["unlabeled"] for r in wire.read_tfrecords(f)]
    assert len(lab) == 5 and len(unl) == 16
    assert [s.root_node.node_id for s in lab] == [0, 1, 2, 3, 4]


def test_link_prediction_on_a_directed_graph_skips_anchors_without_in_edges(workdir):
    """directed graph: an anchor with out-edges (positives exist) but no in-edge has no rooted subgraph in the
    reference's subgraphVIEW, and the INNER JOIN of NodeAnchorBasedLinkPredictionTask.scala:186-194 drops it — no main
    sample for it, while its RootedNodeNeighborhood (random-negative stream) is still written"""
    import yaml
    from gigl_amd.subgraph_sampler import SubgraphSampler
    doc = yaml.safe_load(open(os.path.join(workdir, "configs/nablp_frozen_gbml_config.yaml")))
    doc["sharedConfig"]["isGraphDirected"] = True
    flat = doc["sharedConfig"]["flattenedGraphMetadata"]["nodeAnchorBasedLinkPredictionOutput"]
    flat["tfrecordUriPrefix"] = "out/nablp_dir/main/"
    flat["nodeTypeToRandomNegativeTfrecordUriPrefix"] = {k: "out/nablp_dir/rn/" + k + "/"
                                                         for k in flat["nodeTypeToRandomNegativeTfrecordUriPrefix"]}
    yaml.safe_dump(doc, open(os.path.join(workdir, "configs/nablp_dir_gbml_config.yaml"), "w"))
    SubgraphSampler().run("job", "configs/nablp_dir_gbml_config.yaml", None, uri_base=workdir)
    cfg = GbmlConfigPbWrapper.from_uri("configs/nablp_dir_gbml_config.yaml", uri_base=workdir)
    smp = [wire.NodeAnchorBasedLinkPredictionSample.FromString(r) for f in tfrecord_files(cfg.nablp_tfrecord_uri_prefix)
           for r in wire.read_tfrecords(f)]
    rn = [wire.RootedNodeNeighborhood.FromString(r) for p in cfg.random_negative_tfrecord_uri_prefixes.values()
          for f in tfrecord_files(p) for r in wire.read_tfrecords(f)]
    edge_rows = [wire.decode_tf_example(r) for f in tfrecord_files(os.path.join(
        workdir, "ref_assets/subgraph_sampler/node_anchor_based_link_prediction/edge_data/")) for r in wire.read_tfrecords(f)]
    edges = {(int(np.ravel(r["src"])[0]), int(np.ravel(r["dst"])[0])) for r in edge_rows}
    has_in, has_out = {d for _, d in edges}, {s for s, _ in edges}
    only_out = has_out - has_in
    assert only_out, "the fixture has nodes with out-edges only when read as a directed graph"
    anchors = {s.root_node.node_id for s in smp}
    assert anchors == (has_out & has_in)                 # a positive AND a neighbourhood of its own
    assert not (anchors & only_out)
    assert len(rn) == 27                                  # every node still gets its RootedNodeNeighborhood
    for s in smp:
        for e in s.neighborhood.edges:
            assert (e.src_node_id, e.dst_node_id) in edges   # directed: edges as given, never reversed


def test_subgraph_sampler_on_the_reference_heterogeneous_config(workdir):
    """SubgraphSampler.run on the reference's heterogeneous fixture config (scala/common/src/test/assets/
    subgraph_sampler/heterogeneous/node_anchor_based_link_prediction/frozen_gbml_config_graphdb_dblp_local.yaml, paths
    re-rooted): the typed flow of GraphDBNodeAnchorBasedLinkPredictionTask — one RootedNodeNeighborhood per author and
    per paper under the per-type random-negative prefixes, one NodeAnchorBasedLinkPredictionSample per paper with a
    positive (supervision edge type paper -> author), every record valid against the typed graph"""
    from gigl_amd.subgraph_sampler import SubgraphSampler, load_preprocessed_typed_graph
    uri = "configs/hetero_nablp_frozen_gbml_config.yaml"
    files = SubgraphSampler().run("job", uri, None, uri_base=workdir)
    cfg = GbmlConfigPbWrapper.from_uri(uri, uri_base=workdir)
    assert cfg.is_heterogeneous and cfg.supervision_edge_types == [("paper", "paper_to_author", "author")]
    node_types, num, ids, feats, edges, cet, efeats = load_preprocessed_typed_graph(cfg)
    assert num == {"author": 15, "paper": 19}
    edge_sets = {c: set(zip(edges[et][0].tolist(), edges[et][1].tolist())) for et, c in cet.items()}
    by_cnt = {c: t for t, c in node_types.items()}

    def check_graph(g):
        for x in g.nodes:
            np.testing.assert_array_equal(x.feature_values, feats[by_cnt[x.condensed_node_type]][x.node_id])
        have = {(x.node_id, x.condensed_node_type) for x in g.nodes}
        for e in g.edges:
            assert (e.src_node_id, e.dst_node_id) in edge_sets[e.condensed_edge_type] and e.feature_values.size == 2
            s_t, _, d_t = cfg.condensed_edge_type_map[e.condensed_edge_type]
            assert (e.src_node_id, node_types[s_t]) in have and (e.dst_node_id, node_types[d_t]) in have

    prefixes = cfg.random_negative_tfrecord_uri_prefixes
    for t in ("author", "paper"):
        recs = [wire.RootedNodeNeighborhood.FromString(r) for f in tfrecord_files(prefixes[t])
                for r in wire.read_tfrecords(f)]
        assert sorted(m.root_node.node_id for m in recs) == ids[t].tolist() and files[f"random_negative/{t}"]
        assert all(m.root_node.condensed_node_type == node_types[t] for m in recs)
        for m in recs:
            check_graph(m.neighborhood)
            # two hops of 3 along the edge types that end in the frontier's type: at most 3 + 9 sampled edges
            assert len(m.neighborhood.edges) <= 12
    samples = [wire.NodeAnchorBasedLinkPredictionSample.FromString(r)
               for f in tfrecord_files(cfg.nablp_tfrecord_uri_prefix) for r in wire.read_tfrecords(f)]
    p2a = cet[[et for et in cet if et.relation == "paper_to_author"][0]]
    with_pos = {int(s_) for s_ in set(edges[[et for et in cet if et.relation == "paper_to_author"][0]][0].tolist())}
    assert sorted(m.root_node.node_id for m in samples) == sorted(with_pos)  # isolated roots are dropped
    for m in samples:
        assert m.root_node.condensed_node_type == node_types["paper"] and len(m.pos_edges) == 1  # numPositiveSamples: 1
        e = m.pos_edges[0]
        assert e.src_node_id == m.root_node.node_id and e.condensed_edge_type == p2a
        assert (e.src_node_id, e.dst_node_id) in edge_sets[p2a] and e.feature_values.size == 2
        check_graph(m.neighborhood)
        assert (e.dst_node_id, node_types["author"]) in {(x.node_id, x.condensed_node_type) for x in m.neighborhood.nodes}


def test_typed_sampler_job_options(workdir):
    """numMaxTrainingSamplesToOutput caps the typed training samples; shouldIncludeIsolatedNodesInTraining keeps the roots
    without a positive (empty pos_edges, their own neighbourhood); an explicit SubgraphSamplingStrategy replaces the
    k-hop default DAG (GraphDBNodeAnchorBasedLinkPredictionTask.scala:186-187, 262-283, 428-470)"""
    import yaml
    from gigl_amd.subgraph_sampler import SubgraphSampler
    doc = yaml.safe_load(open(os.path.join(workdir, "configs/hetero_nablp_frozen_gbml_config.yaml")))
    ssc = doc["datasetConfig"]["subgraphSamplerConfig"]
    ssc["numMaxTrainingSamplesToOutput"] = 5
    a2p = {"srcNodeType": "author", "relation": "author_to_paper", "dstNodeType": "paper"}
    p2a = {"srcNodeType": "paper", "relation": "paper_to_author", "dstNodeType": "author"}
    ssc["subgraphSamplingStrategy"] = {"messagePassingPaths": {"paths": [
        {"rootNodeType": "paper", "samplingOps": [
            {"opName": "one", "edgeType": a2p, "randomUniform": {"numNodesToSample": 2}}]},
        {"rootNodeType": "author", "samplingOps": [
            {"opName": "one", "edgeType": p2a, "randomUniform": {"numNodesToSample": 1}}]}]}}
    for k, v in doc["sharedConfig"]["flattenedGraphMetadata"]["nodeAnchorBasedLinkPredictionOutput"].items():
        if isinstance(v, str):
            doc["sharedConfig"]["flattenedGraphMetadata"]["nodeAnchorBasedLinkPredictionOutput"][k] = v.replace("hetero_nablp", "hetero_opts")
        else:
            for t in v:
                v[t] = v[t].replace("hetero_nablp", "hetero_opts")
    uri = "configs/hetero_opts_gbml_config.yaml"
    yaml.safe_dump(doc, open(os.path.join(workdir, uri), "w"))
    SubgraphSampler().run("job", uri, None, uri_base=workdir)
    cfg = GbmlConfigPbWrapper.from_uri(uri, uri_base=workdir)
    read = lambda: [wire.NodeAnchorBasedLinkPredictionSample.FromString(r)
                    for f in tfrecord_files(cfg.nablp_tfrecord_uri_prefix) for r in wire.read_tfrecords(f)]
    capped = read()
    assert 0 < len(capped) <= 5 and all(len(m.pos_edges) == 1 for m in capped)
    # one hop of 2 for the paper + one hop of 1 for its positive author: at most 3 neighbourhood edges
    assert all(len(m.neighborhood.edges) <= 3 for m in capped)
    rn = [wire.RootedNodeNeighborhood.FromString(r) for f in tfrecord_files(cfg.random_negative_tfrecord_uri_prefixes["author"])
          for r in wire.read_tfrecords(f)]
    assert len(rn) == 15 and all(len(m.neighborhood.edges) <= 1 for m in rn)
    ssc["numMaxTrainingSamplesToOutput"] = 0
    doc["sharedConfig"]["shouldIncludeIsolatedNodesInTraining"] = True
    yaml.safe_dump(doc, open(os.path.join(workdir, uri), "w"))
    SubgraphSampler().run("job", uri, None, uri_base=workdir)
    everyone = read()
    assert len(everyone) == 19 and any(len(m.pos_edges) == 0 for m in everyone)
    for m in everyone:
        if not m.pos_edges:
            assert (m.root_node.node_id, 1) in {(x.node_id, x.condensed_node_type) for x in m.neighborhood.nodes}


@pytest.mark.gpu
def test_batched_decoder_and_loss():
    """gigl_linear_batched + gigl_retrieval_loss_batched over G batches == the per-batch entry points, bit for bit
    (same kernels, batch in grid.y); the per-batch loss is itself pinned on the reference's known answers in
    test_link_prediction.py"""
    import torch
    from gigl_amd.engine import default_engine
    from gigl_amd.link_prediction import DecoderType, LinkPredictionDecoder, RetrievalLoss
    dev = torch.device("cuda:0")
    eng = default_engine(dev)
    gen = torch.Generator().manual_seed(5)
    for G, B, n_neg, D in ((3, 40, 70, 64), (5, 130, 1, 128), (1, 7, 9, 32)):
        main = torch.randn(G, 2 * B, D, generator=gen).to(dev)
        rn = torch.randn(G, n_neg, D, generator=gen).to(dev)
        a = torch.randint(0, 50, (G, B), generator=gen).to(dev)  # (small id range: both masks fire)
        cid = torch.randint(0, 50, (G, B + n_neg), generator=gen).to(dev)
        scores = eng.linear_batched(main[:, :B], torch.cat([main[:, B:], rn], dim=1))
        got = eng.retrieval_loss_batched(scores, 0.07, None, a, cid)
        dec = LinkPredictionDecoder(DecoderType.inner_product)
        dec.engine = eng
        loss_fn = RetrievalLoss(temperature=0.07, remove_accidental_hits=True)
        for g in range(G):
            s = dec(main[g, :B], torch.cat([main[g, B:], rn[g]]))
            assert torch.equal(s, scores[:, g])
            want = loss_fn.calculate_batch_retrieval_loss(s, query_ids=a[g], candidate_ids=cid[g])
            assert torch.equal(want, got[g]), (G, B, g, float(want), float(got[g]))
        # with the sampling-probability correction
        prob = torch.rand(G, B + n_neg, generator=gen).to(dev)
        got = eng.retrieval_loss_batched(scores, None, prob, a, None)
        loss_fn = RetrievalLoss()
        for g in range(G):
            want = loss_fn.calculate_batch_retrieval_loss(scores[:, g].contiguous(), candidate_sampling_probability=prob[g],
                                                          query_ids=a[g])
            assert torch.equal(want, got[g])


@pytest.mark.gpu
def test_bench_emulated_world_line(tmp_path):
    """bench.py --workload mag240m-sharded --emulate-world 4 at a toy scale: the measured per-rank quantities and the
    labelled projection are all there, hub-row replication takes rows off the (emulated) links"""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    cp = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "mag240m-sharded", "--emulate-world",
                         "4", "--shard-scale", "0.001", "--batch", "128", "--shard-group", "4", "--steps", "16",
                         "--shard-hot-frac", "0.02"], capture_output=True, text=True, timeout=600,
                        env=dict(os.environ, GIGL_BENCH_CHILD="1"))
    assert cp.returncode == 0, cp.stderr[-2000:]
    line = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["emulated_world"] == 4 and "PROJECTION" in line["value_is"].upper()
    emu = line["emulated"]
    for tag in ("hot_rows", "no_replication"):
        e = emu[tag]
        assert len(e["pulled_rows_per_step_per_rank"]) == 4 and e["wall_ms_per_step_per_rank"] > 0
        # the ranks' own kernel time (HIP events), with the sharded-only groups named, and a roofline on the record
        assert 0 < e["kernel_ms_per_step_per_rank"] <= e["wall_ms_per_step_per_rank"] * 1.05
        assert e["kernel_ms_by_group"]["dist_prep"] > 0 and e["kernel_ms_by_group"]["dist_serve"] > 0
        assert 0 < e["sharded_only_kernel_share"] < 1 and e["roofline"]["kernel"] == "gather_mean" and e["roofline"]["frac"] > 0
        assert 0.0 < e["row_bucket_fill"] <= 1.0 and e["projection"]["label"].startswith("PROJECTION")
    assert line["roofline"]["bound"] == "hbm"
    assert emu["hot_row_hit_rate"]["pulled_rows_with"] < emu["hot_row_hit_rate"]["pulled_rows_without"]
    # the peer-mapped route beside it: no owner-side gather, rows counted per occurrence, and for both routes the OVERLAPPED
    # per-rank step (worlds in flight replayed as hipGraphs) with the link time beside it, hidden and not hidden
    pe = emu["peer_hot_rows"]
    assert pe["route"] == "peer" and "dist_serve" not in pe["kernel_ms_by_group"] and pe["kernel_ms_by_group"]["dist_prep"] > 0
    assert pe["pulled_rows_per_step_mean"] >= emu["hot_rows"]["pulled_rows_per_step_mean"] > 0
    pa = emu["peer_all_hot_rows"]  # ... and the route without any exchange: the hops peer-sampled too
    assert pa["route"] == "peer-all" and pa["kernel_ms_by_group"].get("dist_prep", 0.0) < pe["kernel_ms_by_group"]["dist_prep"]
    for e in (pe, pa, emu["hot_rows"]):
        pj = e["projection"]
        assert e["overlapped"]["ms_per_rank_step"] > 0 and pj["overlapped_step_ms"] == e["overlapped"]["ms_per_rank_step"]
        assert 0 < pj["whole_node_edges_per_s_overlapped_links_not_hidden"] < pj["whole_node_edges_per_s_overlapped_links_hidden"]
    assert line["route"] in ("peer", "peer-all", "bucketed")
    # ... and the same per-rank workload at world 1, measured the same way: the projection as a multiple of it
    w1 = line["world1_reference"]
    assert w1["ms_per_step_overlapped"] > 0 and w1["sampled_plus_aggregated_edges_per_step"] > 0
    assert pe["projection"]["scaling_1_to_4_links_hidden"] > pe["projection"]["scaling_1_to_4_links_not_hidden"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("driver", ["plan", "autograd"])
def test_bench_gat_lp_train_line(tmp_path, driver):
    """bench.py --workload gat-lp --train at a toy scale: the link-prediction TRAINING step of the GAT encoder on the
    in-HBM route (backward + Adam included) emits its line, the loss stays finite and falls from its first value — as the
    library plan (gigl_gat_nablp_train_plan_*, the default) and as the autograd-driven loop (--gat-train-autograd)"""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    cp = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "gat-lp", "--train", "--shard-scale",
                         "0.001", "--batch", "64", "--steps", "16", "--min-seconds", "0.2"]
                        + (["--gat-train-autograd"] if driver == "autograd" else []), capture_output=True,
                        text=True, timeout=600, env=dict(os.environ, GIGL_BENCH_CHILD="1"))
    assert cp.returncode == 0, cp.stderr[-2000:]
    line = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["metric"] == "sampled+aggregated edges/s" and line["value"] > 0 and "TRAINING" in line["config"]["workload"]
    if driver == "plan":
        assert "gigl_gat_nablp_train_plan" in line["config"]["driver"]
        assert line["config"]["loss_last_step"] < line["config"]["loss_first_step"]
        assert line["config"]["autograd_driven_ms_per_step"] > 0  # (the comparison ran)
        return
    assert line["config"]["loss_last_mean"] < line["config"]["loss_first"]
    assert "gather_mean" in line["roofline"]["by_kernel"] and "linear" in line["roofline"]["by_kernel"]
