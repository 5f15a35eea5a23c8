"""debug: the two-rank inferencer on the overflow job, every staged batch checked against the oracle's union graph"""
import sys, os, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch


def worker(rank, world, port, base, cfg_uri, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port), GIGL_DIST_BACKEND="gloo")
        import oracle
        from gigl_amd import config, hbm
        from gigl_amd.inferencer import Inferencer
        from test_gpu_overflow import N, E, FAN
        from helpers import rmat_edges
        src, dst = rmat_edges(15, E, 7)
        src, dst = (src.astype(np.int64) * 0x9E3779B1) % N, (dst.astype(np.int64) * 0x9E3779B1) % N
        keep = src != dst
        rowptr, col = oracle.build_csc(N, src[keep].astype(np.uint32), dst[keep].astype(np.uint32), is_directed=False)
        x = np.random.default_rng(7).standard_normal((N, 16)).astype(np.float32)
        orig = hbm.ResidentGraph.graph_data
        log = []

        def patched(self, roots, pad_to=None, wide=False):
            g, ri = orig(self, roots, pad_to, wide)
            r_h = roots.cpu().numpy().view(np.uint32)
            nbr, _ = oracle.sample_khop(rowptr, col, r_h, list(FAN), canonical=True)
            u = oracle.union_build(r_h, list(FAN), nbr)
            n = int(u["meta"][0])
            ok_n = g.x.shape[0] == n
            ok_x = ok_n and np.array_equal(g.x.cpu().numpy(), x[u["nodes"][:n].astype(np.int64)])
            ok_rp = ok_n and np.array_equal(g.rowptr.cpu().numpy(), u["rowptr"][:n + 1])
            ok_ri = np.array_equal(ri.cpu().numpy(), u["root_local"])
            log.append((wide, ok_n, ok_x, ok_rp, ok_ri, int(g.x.shape[0]), n))
            return g, ri
        hbm.ResidentGraph.graph_data = patched
        inf = Inferencer()
        out = inf.run("job", cfg_uri, None, uri_base=base, route="hbm")
        q.put((rank, "ok", out, log))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e), []))


if __name__ == "__main__":
    import torch.multiprocessing as mp
    from test_gpu_hbm_route import _write_small_job, _variant, _rows
    from test_gpu_overflow import N, E, FAN, B
    from gigl_amd.models import GraphSAGE
    base = tempfile.mkdtemp()
    n, src, dst, x = _write_small_job(base, n=N, e=E, d=16, hid=32, out_dim=8, fan=FAN, batch=B)
    torch.manual_seed(5)
    model = GraphSAGE(16, 32, 8, num_layers=2)
    os.makedirs(os.path.join(base, "out/model"), exist_ok=True)
    torch.save(model.state_dict(), os.path.join(base, "out/model/model.pt"))
    cfg2 = _variant(base, "configs/job.yaml", "w2")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29911, base, cfg2, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for rank, status, out, log in res:
        print("rank", rank, status, out if status != "ok" else "")
        for l in log:
            print("   staged batch: wide=%s n_ok=%s x_ok=%s rowptr_ok=%s root_local_ok=%s n=%d/%d" % l)
    import oracle
    from oracle import gnn_ref
    from gigl_amd.inferencer import Inferencer
    rowptr, col = oracle.build_csc(n, src.astype(np.uint32), dst.astype(np.uint32), is_directed=False)
    sd = torch.load(os.path.join(base, "out/model/model.pt"), map_location="cpu")

    def check(tag, path):
        rows = _rows(path)
        for lo in range(0, min(len(rows), 4 * B), B):
            roots = np.array([r["node_id"] for r in rows[lo:lo + B]], dtype=np.uint32)
            got = np.array([r["emb"] for r in rows[lo:lo + B]], np.float32)
            nbr, _ = oracle.sample_khop(rowptr, col, roots, list(FAN), canonical=True)
            u = oracle.union_build(roots, list(FAN), nbr)
            ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
            want = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"].astype(np.int64)]), ei, sd, 2)[u["root_local"]].numpy()
            print(f"{tag}: batch at {lo}: roots {roots[:4]}... max err vs oracle {np.abs(got - want).max():.3g}")
    for rank, status, out, log in res:
        if status == "ok":
            check(f"rank {rank}", out["embeddings"])
    single = Inferencer().run("job", _variant(base, "configs/job.yaml", "w1"), None, uri_base=base, route="hbm")
    check("single", single["embeddings"])
