"""EmbeddingExporter on the device (gigl_avro_embeddings_encode) against the Avro oracle and the expectations of the
reference's own unit test (python/tests/unit/common/data/export_test.py)."""
import numpy as np
import pytest
import torch

from oracle import avro
from gigl_amd import export
from gigl_amd.engine import HipEngine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = HipEngine(0)
    yield e
    e.close()


def _layout(eng, n, d, tlen):
    import ctypes as C
    per, nb, cap = C.c_int32(), C.c_int64(), C.c_int64()
    assert eng._lib.gigl_avro_embeddings_layout(n, d, tlen, C.byref(per), C.byref(nb), C.byref(cap)) == 0
    return per.value, nb.value, cap.value


@pytest.mark.parametrize("n,d,ty", [(1, 1, "u"), (7, 3, "user"), (300, 128, "paper"), (5000, 16, "item"),
                                    (40, 0, "empty"), (9, 1000, "wide"), (4100, 1, ""), (65, 5, "ünïcode-节点")])
def test_blocks_are_byte_identical_to_the_oracle(eng, n, d, ty):
    g = torch.Generator().manual_seed(n * 31 + d)
    # ids across every varint length, negative ids included (Avro long is signed)
    mags = torch.randint(0, 63, (n,), generator=g)
    ids = (torch.randint(0, 2**62, (n,), generator=g) >> (62 - mags)) * (torch.randint(0, 2, (n,), generator=g) * 2 - 1)
    ids[0] = 0
    emb = torch.randn(n, d, generator=g)
    if d:
        emb[0, 0] = float("nan")
        emb[-1, -1] = float("-inf")
    sync = bytes(range(100, 116))
    blocks, rec_off = eng.encode_avro_embeddings(ids, emb, ty, sync)
    per, nb, cap = _layout(eng, n, d, len(ty.encode()))
    want = avro.encode_embedding_blocks(ids.tolist(), emb.numpy(), ty, sync, per)
    got = blocks.cpu().numpy().tobytes()
    assert len(got) <= cap and got == want
    off = rec_off.cpu().numpy()
    for i in (0, n // 2, n - 1):
        rec = avro.encode_embedding_record(int(ids[i]), ty, emb[i].numpy())
        assert got[off[i]:off[i] + len(rec)] == rec


def test_strided_and_low_precision_embeddings(eng):
    base = torch.randn(50, 64, device=eng.device)
    view = base[:, 8:40]  # row stride 64, 32 columns
    ids = torch.arange(50, dtype=torch.int32)
    sync = b"s" * 16
    blocks, _ = eng.encode_avro_embeddings(ids, view, "t", sync)
    per, _, _ = _layout(eng, 50, 32, 1)
    assert blocks.cpu().numpy().tobytes() == avro.encode_embedding_blocks(ids.tolist(), view.cpu().numpy(), "t", sync, per)
    half = base[:, :16].to(torch.bfloat16)
    blocks, _ = eng.encode_avro_embeddings(ids, half, "t", sync)
    per, _, _ = _layout(eng, 50, 16, 1)
    assert blocks.cpu().numpy().tobytes() == avro.encode_embedding_blocks(ids.tolist(), half.float().cpu().numpy(), "t",
                                                                          sync, per)


def test_reference_unit_test_expectations(eng, tmp_path):
    # export_test.py:63-111 (one shard, two batches, integer embeddings), :113-172 (context exit then explicit flush),
    # :175-228 (threshold flush per batch), prefix naming
    id_batches = [torch.tensor([1, 2, 3]), torch.tensor([4, 5, 6])]
    emb_batches = [torch.tensor([[1, 11], [2, 12], [3, 13]]), torch.tensor([[4, 14], [5, 15], [6, 16]])]
    expect = [{"node_id": k, "node_type": "test_type", "emb": [float(k), float(k + 10)]} for k in range(1, 7)]
    d1 = tmp_path / "one"
    with export.EmbeddingExporter(str(d1), file_prefix="my-prefix", engine=eng) as ex:
        for i, e in zip(id_batches, emb_batches):
            ex.add_embedding(i, e, "test_type")
    assert [p.name for p in d1.iterdir()] == ["my-prefix_00000000.avro"]
    schema, recs = avro.read_embedding_file((d1 / "my-prefix_00000000.avro").read_bytes())
    assert schema == export.AVRO_SCHEMA and recs == expect
    d2 = tmp_path / "two"
    ex = export.EmbeddingExporter(str(d2), engine=eng)
    with ex:
        ex.add_embedding(id_batches[0], emb_batches[0], "test_type")
    ex.add_embedding(id_batches[1], emb_batches[1], "test_type")
    ex.flush_embeddings()
    ex.flush_embeddings()  # empty: skipped
    assert sorted(p.name for p in d2.iterdir()) == ["shard_00000000.avro", "shard_00000001.avro"]
    assert avro.read_embedding_file((d2 / "shard_00000000.avro").read_bytes())[1] == expect[:3]
    assert avro.read_embedding_file((d2 / "shard_00000001.avro").read_bytes())[1] == expect[3:]
    d3 = tmp_path / "three"
    with export.EmbeddingExporter(str(d3), min_shard_size_threshold_bytes=1, engine=eng) as ex:
        for i, e in zip(id_batches, emb_batches):
            ex.add_embedding(i, e, "test_type")
    assert sorted(p.name for p in d3.iterdir()) == ["shard_00000000.avro", "shard_00000001.avro"]
    assert avro.read_embedding_file((d3 / "shard_00000001.avro").read_bytes())[1] == expect[3:]


def test_full_size_export_checksum(eng, tmp_path):
    # products-sized output: 2,449,029 embeddings of 128 floats; the shard is decoded with numpy by the fixed record
    # layout (all ids < 2^31 here so only the id varint length varies) and compared by content
    n, d = 2_449_029, 128
    g = torch.Generator(device=eng.device).manual_seed(0)
    emb = torch.randn(n, d, device=eng.device, generator=g)
    ids = torch.randperm(n, device=eng.device, generator=g)
    sync = bytes(range(16))
    blocks, rec_off = eng.encode_avro_embeddings(ids, emb, "user", sync)
    per, nb, cap = _layout(eng, n, d, 4)
    data = blocks.cpu().numpy()
    off = rec_off.cpu().numpy()
    assert np.all(np.diff(off) > 0)
    ids_h = ids.cpu().numpy()
    zz = ids_h.astype(np.uint64) << np.uint64(1)
    il = np.ones(n, dtype=np.int64)
    for k in range(1, 5):
        il += (zz >= (1 << (7 * k))).astype(np.int64)
    hl = il + 1 + 4 + 2  # id | len("user") "user" | count 128 -> 2 bytes
    # payload bytes of every record, gathered through the returned offsets
    rows = data[(off + hl)[:, None] + np.arange(4 * d)[None, :]].view("<f4")
    assert np.array_equal(rows.view(np.uint32), emb.cpu().numpy().view(np.uint32))
    assert np.all(data[off + hl + 4 * d] == 0)
    # ids decoded from the varints
    dec = np.zeros(n, dtype=np.uint64)
    for k in range(5):
        b = data[off + np.minimum(k, il - 1)].astype(np.uint64)
        dec |= np.where(k < il, (b & np.uint64(0x7F)) << np.uint64(7 * k), np.uint64(0))
    assert np.array_equal(dec >> np.uint64(1), ids_h.astype(np.uint64))
    # block framing: walk every block
    assert nb == (n + per - 1) // per
    r = avro._Reader(data.tobytes())
    seen = 0
    while not r.eof():
        cnt, size = r.long(), r.long()
        assert off[seen] == r.p
        r.p += size
        assert bytes(data[r.p:r.p + 16]) == sync
        r.p += 16
        seen += cnt
    assert seen == n
