"""The sampler's 32-bit-proxy fast path must fall back to the exact 64-bit selection whenever a proxy tie could
matter.  GIGL_SAMPLER_PROXY_BITS (test knob) keeps only the top N bits of the proxy, so ties become frequent
(N=4: 16 distinct keys) and the fallback decides almost every row — results must still equal the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle
from helpers import rmat_edges
from gigl_amd.engine import HipEngine
s, d = rmat_edges(14, 700000, seed=5)
n = 1 << 14
rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
assert np.diff(rowptr).max() > 2000
eng = HipEngine(0)
eng.load_csc(rowptr, col)
roots = np.random.default_rng(0).integers(0, n, size=600).astype(np.uint32)
for fan in ([25, 10], [64, 2]):
    t = eng.sample_khop(roots, fan)
    nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots, fan, canonical=True)
    for k in range(2):
        assert np.array_equal(t.cnt[k].cpu().numpy(), cnt_o[k])
        assert np.array_equal(t.nbr[k].cpu().numpy().view(np.uint32), nbr_o[k]), (fan, k)
print("OK")
"""


@pytest.mark.parametrize("bits", ["4", "9", "20"])
def test_forced_proxy_ties_fall_back_to_exact(bits):
    env = dict(os.environ, GIGL_SAMPLER_PROXY_BITS=bits)
    out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "tests"))], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]
