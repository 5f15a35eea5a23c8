"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box
(MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern").

  1. streaming copy of a 2 GiB fp32 tensor (torch elementwise copy kernel): reads 2 GiB, writes 2 GiB
  2. gigl gather_rows of R random 100-float rows out of a 2.4 M-row table (the feature-gather pattern):
     algorithmic read = R*(400 + 4) B, write = R*400 B
Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and again with WRITE_SIZE); scripts/pmc_summary.py turns
the per-dispatch counters into per-kernel averages and the correction factors."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from gigl_amd.engine import HipEngine  # noqa: E402


def main():
    eng = HipEngine(0)
    dev = eng.device
    n_copy = (2 << 30) // 4
    a = torch.empty(n_copy, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    torch.cuda.synchronize()
    for _ in range(4):
        torch.add(a, 1.0, out=b)  # an elementwise kernel (a same-dtype copy_ would be a runtime memcpy)
    torch.cuda.synchronize()
    n, d, r = 2_400_000, 100, 1 << 20
    x = torch.randn(n, d, device=dev)
    eng.load_features(x)
    idx = torch.from_numpy(np.random.default_rng(0).integers(0, n, size=r).astype(np.int32)).to(dev)
    n_dev = torch.tensor([r], dtype=torch.int32, device=dev)
    for _ in range(4):
        out = eng.gather_rows(idx, n_dev, r)
    torch.cuda.synchronize()
    assert torch.equal(out[:1000], x[idx[:1000].long()])
    print(json.dumps({"copy_read_bytes": n_copy * 4, "copy_write_bytes": n_copy * 4,
                      "gather_rows_read_bytes": r * (d * 4 + 4), "gather_rows_write_bytes": r * d * 4}))
    eng.close()


if __name__ == "__main__":
    main()
