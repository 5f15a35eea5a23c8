"""The LDS-staged two-hop union build of the one-call plans ("LG3", csrc/union.hip) against the HBM-table build it
replaces ("LG2", kept behind GIGL_UNION_LG2=1): every output of the plan's union graph — node list, level counts, rows
(after their sort: ascending, duplicate-free), root_local, edge count — and the plan's root rows must be identical bit
for bit, on batches that exercise roots that are each other's neighbours (extras), repeated roots, several batches per
call, hubs, multi-edge graphs and invalid roots.  (LG2 itself is checked against the generic build and the oracle in
test_gpu_plan.py / test_gpu_groups.py; this file pins that nothing moved when the table went to LDS.)

Reference semantics kept: /root/reference/python/gigl/src/common/graph_builder/abstract_graph_builder.py:16-24,100-150
(first-seen numbering per level, edges deduplicated per batch)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")

_WORKER = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import oracle
from helpers import rmat_edges
from gigl_amd.engine import HipEngine
from gigl_amd.models import GraphSAGE
out = {}
def case(name, n, src, dst, directed, multi, b, fan, groups, roots_fn, d=16):
    eng = HipEngine(0)
    eng.build_from_coo(n, src.astype(np.uint32), dst.astype(np.uint32), is_directed=directed, keep_multi_edges=multi)
    eng.load_features(np.random.default_rng(1).standard_normal((n, d)).astype(np.float32))
    torch.manual_seed(0)
    model = GraphSAGE(d, 32, 8, num_layers=2).to(eng.device)
    plan = model.make_plan(eng, b, fan, groups=groups)
    for it in range(2):
        roots = roots_fn(it).astype(np.uint32)
        rows = plan.run(torch.from_numpy(roots.view(np.int32)).to(eng.device)).cpu().numpy()
        hb = plan.last_batch_to_host()
        nn = int(hb["meta"][0])
        out[f"{name}/{it}/rows"] = rows
        out[f"{name}/{it}/meta"] = hb["meta"]
        out[f"{name}/{it}/nodes"] = hb["nodes"]
        out[f"{name}/{it}/root_local"] = hb["root_local"]
        # rows in local-id order (where a row sits in col is free)
        lens = (hb["rowend"][:nn] - hb["rowptr"][:nn]).astype(np.int64)
        out[f"{name}/{it}/lens"] = lens
        out[f"{name}/{it}/cols"] = np.concatenate([hb["col"][s:s + l] for s, l in zip(hb["rowptr"][:nn], lens)] + [np.zeros(0, np.int32)])
    plan.close(); eng.close()

rng = np.random.default_rng(5)
s, d_ = rmat_edges(13, 150000, seed=8); n = 1 << 13
case("rmat", n, s, d_, False, False, 256, [25, 10], 1, lambda it: np.random.default_rng(it).integers(0, n, 256))
case("rmat_groups", n, s, d_, False, False, 64, [25, 10], 4, lambda it: np.random.default_rng(10 + it).integers(0, n, 256))
case("rmat_15_10_b300", n, s, d_, False, False, 300, [15, 10], 1, lambda it: np.random.default_rng(20 + it).integers(0, n, 300))
# a small dense graph: most hop-0 nodes are roots themselves (extras everywhere), repeated roots, an invalid root
ns = 300
ss, ds = rng.integers(0, ns, 4000), rng.integers(0, ns, 4000)
def dense_roots(it):
    r = np.random.default_rng(30 + it).integers(0, ns, 128).astype(np.int64)
    r[5] = r[4]; r[77] = r[0]
    if it == 1: r[9] = 0xFFFFFFFF
    return r
case("dense", ns, ss, ds, False, False, 128, [10, 5], 1, dense_roots)
case("dense_groups", ns, ss, ds, False, False, 32, [10, 5], 4, dense_roots)
case("dense_all_roots", ns, ss, ds, False, False, 300, [10, 5], 1, lambda it: np.arange(300))
case("dense_directed_multi", ns, np.concatenate([ss, ss[:500]]), np.concatenate([ds, ds[:500]]), True, True, 128, [10, 5], 1, dense_roots)
# a hub: every node points at node 0, node 0 is a root and everybody's neighbour
hs = np.concatenate([np.arange(1, 2000), rng.integers(0, 2000, 6000)]); hd = np.concatenate([np.zeros(1999, np.int64), rng.integers(0, 2000, 6000)])
case("hub", 2000, hs, hd, False, False, 64, [25, 10], 2, lambda it: np.concatenate([[0], np.random.default_rng(40 + it).integers(0, 2000, 127)]))
np.savez(sys.argv[2], **out)
'''


@pytest.mark.gpu
def test_lds_staged_union_equals_the_hbm_table_build(tmp_path):
    res = {}
    # (GIGL_LG3_MIN_WGS=0: calls of a few batches would otherwise be routed to LG2 — they finish sooner there)
    for tag, env in (("lg3", {"GIGL_LG3_MIN_WGS": "0"}), ("lg2", {"GIGL_UNION_LG2": "1"})):
        path = str(tmp_path / f"{tag}.npz")
        e = dict(os.environ, **env)
        e.pop("GIGL_UNION_GENERIC", None)
        subprocess.run([sys.executable, "-c", _WORKER, os.path.abspath(ROOT), path], check=True, env=e, timeout=900)
        res[tag] = np.load(path)
    assert sorted(res["lg3"].files) == sorted(res["lg2"].files) and len(res["lg3"].files) > 40
    for k in res["lg2"].files:
        a, b = res["lg3"][k], res["lg2"][k]
        assert a.shape == b.shape, k
        if k.endswith("/rows"):
            assert np.array_equal(a, b, equal_nan=True), k
        else:
            assert np.array_equal(a, b), k
    # the cases exercised what they were built for: extras and overflow-free numbering
    m = res["lg3"]["dense/0/meta"]
    assert m[8] == 0 and m[2] < 128 and m[3] > m[2]
