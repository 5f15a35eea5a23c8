#!/usr/bin/env python3
"""What the hardware does with the record encoder's payload pattern in isolation (scripts/micro/rowcopy.hip, built by
`hipcc -shared` into scripts/micro/librowcopy.so): 368,640 random 400-byte rows of a 2.45 M-row table copied to
byte-misaligned destinations 410 bytes apart.  Prints us per launch and TB/s (read + written)."""
import ctypes as C
import os
import torch

import subprocess

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "micro", "librowcopy.so")
if not os.path.exists(so):  # (built artefacts are not in the history)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared",
                           os.path.join(here, "micro", "rowcopy.hip"), "-o", so])
lib = C.CDLL(so)
lib.rowcopy_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int,
                               C.c_int, C.c_void_p]
n, d, rows = 2_449_029, 100, 4096 * 90
g = torch.Generator(device="cuda").manual_seed(1)
feat = torch.randn(n, d, device="cuda")
ids = torch.randint(0, n, (rows,), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
off = (torch.arange(rows, device="cuda", dtype=torch.int64) * 410 + 13)
out = torch.zeros(rows * 410 + 4096, dtype=torch.uint8, device="cuda")
for variant, name in ((0, "copy, unaligned 16-byte stores"), (2, "row loads only"), (3, "stores only")):
    for blocks, unr in ((2048, 4), (2048, 8), (4096, 4), (8192, 4), (1024, 8)):
        def run():
            rc = lib.rowcopy_launch(feat.data_ptr(), d, ids.data_ptr(), off.data_ptr(), rows, out.data_ptr(), variant,
                                    blocks, unr, None)
            assert rc == 0
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        moved = rows * 400 * (2 if variant == 0 else 1)
        print(f"{name:32s} blocks {blocks:5d} unroll {unr}: {us:7.1f} us  {moved/us/1e6:6.2f} TB/s")
got = out.view(-1)
chk = torch.stack([got[13 + 410 * i: 13 + 410 * i + 400] for i in (0, 5, rows - 1)])
ref = feat[ids[[0, 5, rows - 1]].long()].view(torch.uint8).view(3, 400)
print("rows correct:", bool(torch.equal(chk, ref)))
