"""oracle/avro.py pinned on the Apache Avro specification's published examples, plus the host-side behaviour of
gigl_amd.export.EmbeddingExporter that needs no GPU (argument checks, header layout)."""
import json

import pytest

from oracle import avro
from gigl_amd import export


def test_zigzag_table_of_the_specification():
    # Avro spec, "Binary Encoding / Primitive Types": value -> hex
    table = {0: "00", -1: "01", 1: "02", -2: "03", 2: "04", -64: "7f", 64: "8001", 8192: "808001", -8193: "818001"}
    for v, hx in table.items():
        assert avro.encode_long(v).hex() == hx
        assert export._avro_long(v).hex() == hx
        assert avro._Reader(bytes.fromhex(hx)).long() == v
    assert avro.encode_long(2**63 - 1).hex() == "feffffffffffffffff01"
    assert avro.encode_long(-2**63).hex() == "ffffffffffffffffff01"
    assert avro._Reader(avro.encode_long(-2**63)).long() == -2**63


def test_string_and_array_examples_of_the_specification():
    assert avro.encode_string("foo").hex() == "06666f6f"  # spec: 06 66 6f 6f
    assert avro.encode_array([3, 27], avro.encode_long).hex() == "04063600"  # spec: 04 06 36 00
    assert avro.encode_array([], avro.encode_long).hex() == "00"
    assert avro.encode_float(1.0).hex() == "0000803f"  # little-endian IEEE single


def test_record_layout():
    rec = avro.encode_embedding_record(3, "user", [1.0, 11.0])
    assert rec.hex() == "06" + "08" + b"user".hex() + "04" + "0000803f" + "00003041" + "00"


def test_header_and_file_round_trip_on_the_reference_test_records():
    # python/tests/unit/common/data/export_test.py:63-111: two batches, six records
    sync = bytes(range(16))
    ids = [[1, 2, 3], [4, 5, 6]]
    emb = [[[1, 11], [2, 12], [3, 13]], [[4, 14], [5, 15], [6, 16]]]
    data = export.avro_file_header(sync)
    assert data[:4] == b"Obj\x01" and data[-16:] == sync
    for i, e in zip(ids, emb):
        data += avro.encode_embedding_blocks(i, e, "test_type", sync, records_per_block=2)
    schema, recs = avro.read_embedding_file(data)
    assert schema == export.AVRO_SCHEMA == json.loads(json.dumps(export.AVRO_SCHEMA))
    assert recs == [{"node_id": k, "node_type": "test_type", "emb": [float(k), float(k + 10)]} for k in range(1, 7)]


def test_reader_rejects_a_wrong_sync_marker():
    sync = bytes(16)
    data = export.avro_file_header(sync) + avro.encode_embedding_blocks([1], [[0.5]], "t", b"\x01" * 16, 1)
    with pytest.raises(AssertionError, match="sync"):
        avro.read_embedding_file(data)


def test_exporter_argument_checks(tmp_path):
    # python/tests/unit/common/data/export_test.py:29-48
    with pytest.raises(ValueError, match="file_flush_threshold must be a non-negative integer, but got -1"):
        export.EmbeddingExporter(str(tmp_path), min_shard_size_threshold_bytes=-1)
    with pytest.raises(ValueError, match="local or mounted"):
        export.EmbeddingExporter("gs://test-bucket/test-folder")
    ex = export.EmbeddingExporter(str(tmp_path))
    with ex:
        with pytest.raises(RuntimeError, match="already in a context"):
            with ex:
                pass
    ex.flush_embeddings()  # nothing buffered: no file (export_test.py:297-304)
    assert list(tmp_path.iterdir()) == []
