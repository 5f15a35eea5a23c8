"""Auto-activating pin of the sampler on the REFERENCE's deterministic permutation: when a maintainer has produced
tests/golden/spark_deterministic_sample.json with scripts/spark/deterministic_sample.scala (it needs a JVM, Spark and
the reference's sampler jar — none exist in the build image), the oracle (CPU) and gigl_sample_khop (GPU) must
reproduce every sampled row of it as a set.  Until then both tests skip and the sampler stays "parity unpinned"
(DESIGN.md section 3).  The input edge list the snippet reads is committed and checked here against the fixture."""
import json
import os

import numpy as np
import pytest

import oracle
from helpers import load_fixture_graph

GOLD = "spark_deterministic_sample.json"


def _graph(golden_dir):
    n, src, dst, _ = load_fixture_graph(golden_dir)
    return n, *oracle.build_csc(n, src, dst, is_directed=False)


def test_snippet_input_is_the_bidirectionalised_fixture(golden_dir):
    n, rowptr, col = _graph(golden_dir)
    rows = [l.strip().split(",") for l in open(os.path.join(golden_dir, "spark_input_edges.csv"))][1:]
    got = sorted((int(d), int(s)) for s, d in rows)
    want = sorted((v, int(s)) for v in range(n) for s in col[rowptr[v]:rowptr[v + 1]])
    assert got == want


def _check(golden_dir, sample):
    path = os.path.join(golden_dir, GOLD)
    if not os.path.exists(path):
        pytest.skip(f"{GOLD} has not been generated (needs a JVM + Spark + the reference jar: scripts/spark/)")
    g = json.load(open(path))
    f = int(g["fanout"])
    n, rowptr, col = _graph(golden_dir)
    roots = np.arange(n, dtype=np.uint32)
    nbr = sample(rowptr, col, roots, [f, f], int(g["sampling_seed"]))
    h1 = nbr[0].reshape(n, f)
    h2 = nbr[1].reshape(n, f, f)
    for row in g["hop1"]:
        got = set(int(v) for v in h1[row["root"]] if v != 0xFFFFFFFF)
        assert got == set(row["sampled"]), row
    for row in g["hop2"]:
        r, p = row["root"], row["parent"]
        j = [int(v) for v in h1[r]].index(p)
        got = set(int(v) for v in h2[r, j] if v != 0xFFFFFFFF)
        assert got == set(row["sampled"]), row
    assert len(g["hop1"]) == int((np.diff(rowptr) > 0).sum())


def test_oracle_matches_the_spark_sample(golden_dir):
    _check(golden_dir, lambda rp, cl, roots, fan, seed: oracle.sample_khop(rp, cl, roots, fan, sampling_seed=seed,
                                                                           canonical=True)[0])


@pytest.mark.gpu
def test_device_sampler_matches_the_spark_sample(golden_dir):
    def sample(rp, cl, roots, fan, seed):
        from gigl_amd.engine import HipEngine
        eng = HipEngine(0)
        try:
            eng.load_csc(rp, cl)
            t = eng.sample_khop(roots, fan, sampling_seed=seed)
            return [x.cpu().numpy().view(np.uint32) for x in t.nbr]
        finally:
            eng.close()
    _check(golden_dir, sample)
