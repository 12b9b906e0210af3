#!/usr/bin/env python3
"""Generate tests/golden/test_data_sv.npz from the reference's own test fixture.

Source: /root/reference/pinot-core/src/test/resources/data/test_data-sv.avro (30 000 rows, null codec, no
nulls), the Avro file behind BaseSingleValueQueriesTest (pinot-core/src/test/java/org/apache/pinot/queries/
BaseSingleValueQueriesTest.java:53-106).  /root/reference does not exist on the GPU box, so the columns the
golden queries touch are re-encoded here as a compressed .npz that travels with the repo:
  int columns      -> int32 arrays
  string columns   -> sorted unique values (the dictionary Pinot would build) + int32 dictIds
The expected results live in tests/golden/golden_queries.json, copied by hand from
InnerSegmentAggregationSingleValueQueriesTest.java:44-112 and InterSegmentAggregationSingleValueQueriesTest.java.

Run in the build container only:  python tools/make_golden_fixture.py
"""
import json
import os
import sys

import numpy as np

SRC = "/root/reference/pinot-core/src/test/resources/data/test_data-sv.avro"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "test_data_sv.npz")

# columns of BaseSingleValueQueriesTest.SCHEMA
INT_COLUMNS = ["column1", "column3", "column6", "column7", "column9", "column17", "column18", "daysSinceEpoch"]
STRING_COLUMNS = ["column5", "column11", "column12"]


class AvroReader:
    """Minimal Avro object-container reader: null codec, records of ["null", T] unions, T in {int, string}."""

    def __init__(self, data):
        self.b = data
        self.pos = 0

    def read_long(self):
        shift = 0
        n = 0
        while True:
            c = self.b[self.pos]
            self.pos += 1
            n |= (c & 0x7F) << shift
            if not c & 0x80:
                break
            shift += 7
        return (n >> 1) ^ -(n & 1)

    def read_bytes(self):
        n = self.read_long()
        v = self.b[self.pos:self.pos + n]
        self.pos += n
        return v


def read_avro(path):
    r = AvroReader(open(path, "rb").read())
    assert r.b[:4] == b"Obj\x01"
    r.pos = 4
    meta = {}
    while True:
        cnt = r.read_long()
        if cnt == 0:
            break
        if cnt < 0:
            r.read_long()
            cnt = -cnt
        for _ in range(cnt):
            k = r.read_bytes()
            meta[k] = r.read_bytes()
    assert meta.get(b"avro.codec", b"null") == b"null"
    schema = json.loads(meta[b"avro.schema"])
    fields = [(f["name"], [t for t in f["type"] if t != "null"][0], f["type"].index("null")) for f in schema["fields"]]
    sync = r.b[r.pos:r.pos + 16]
    r.pos += 16
    rows = {name: [] for name, _, _ in fields}
    while r.pos < len(r.b):
        nrec = r.read_long()
        r.read_long()  # block byte size
        for _ in range(nrec):
            for name, typ, null_idx in fields:
                branch = r.read_long()
                if branch == null_idx:
                    raise ValueError("unexpected null in fixture")
                rows[name].append(r.read_long() if typ == "int" else r.read_bytes().decode("utf-8"))
        assert r.b[r.pos:r.pos + 16] == sync
        r.pos += 16
    return rows


def main():
    if not os.path.exists(SRC):
        sys.exit("reference fixture not found (this script only runs in the build container)")
    rows = read_avro(SRC)
    n = len(rows["column1"])
    assert n == 30000
    out = {}
    for c in INT_COLUMNS:
        out[c] = np.asarray(rows[c], dtype=np.int32)
    for c in STRING_COLUMNS:
        values = sorted(set(rows[c]))  # Pinot sorts string dictionaries by UTF-8 bytes; ASCII here so str order matches
        index = {v: i for i, v in enumerate(values)}
        out[c + "__dict"] = np.asarray(values, dtype="U")
        out[c + "__ids"] = np.asarray([index[v] for v in rows[c]], dtype=np.int32)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    # sanity: the unfiltered goldens of InnerSegmentAggregationSingleValueQueriesTest.java:50
    assert int(out["column1"].astype(np.int64).sum()) == 32317185437847
    assert int(out["column3"].max()) == 2147419555
    assert int(out["column6"].min()) == 1689277
    assert int(out["column7"].astype(np.int64).sum()) == 28175373944314
    print("unfiltered goldens reproduce")


def export_prebuilt_segment():
    """pinot-core/src/test/resources/data/paddingOld.tar.gz is a 5-doc v1-format segment that was written by the
    reference's own Java writers (FixedBitSVForwardIndexWriter, SegmentDictionaryCreator): real golden BYTES for the
    fixed-bit forward index and the dictionary layouts.  The few bytes are stored as hex next to the metadata keys."""
    import io
    import tarfile
    src = "/root/reference/pinot-core/src/test/resources/data/paddingOld.tar.gz"
    out = {"_source": src, "columns": {}}
    with tarfile.open(src) as tar:
        files = {m.name.split("/", 1)[1]: tar.extractfile(m).read() for m in tar.getmembers() if m.isfile()}
    meta = {}
    for line in files["metadata.properties"].decode().splitlines():
        if "=" in line and not line.startswith("#"):
            k, v = line.split("=", 1)
            meta[k.strip()] = v.strip()
    out["total_docs"] = int(meta["segment.total.docs"])
    for col in ("age", "percent", "outgoingName1", "name"):
        out["columns"][col] = {
            "dataType": meta["column.%s.dataType" % col], "cardinality": int(meta["column.%s.cardinality" % col]),
            "bitsPerElement": int(meta["column.%s.bitsPerElement" % col]), "isSorted": meta["column.%s.isSorted" % col],
            "dict_hex": files[col + ".dict"].hex(), "fwd_hex": files[col + ".sv.unsorted.fwd"].hex()}
    path = os.path.join(os.path.dirname(OUT), "pinot_v1_segment_paddingOld.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


def export_segment_directories():
    """The three complete v1 segment directories under pinot-core/src/test/resources/data/ (paddingOld / paddingPercent /
    paddingNull: 5 docs, INT + FLOAT + LONG + STRING columns, written by the reference's Java segment creator with the three
    string-padding conventions).  Every file is stored as hex so a test can lay the directory out again and open it with the
    native segment loader."""
    import tarfile
    out = {}
    for name in ("paddingOld", "paddingPercent", "paddingNull"):
        src = "/root/reference/pinot-core/src/test/resources/data/%s.tar.gz" % name
        with tarfile.open(src) as tar:
            out[name] = {"_source": src, "files": {m.name.split("/", 1)[1]: tar.extractfile(m).read().hex() for m in tar.getmembers() if m.isfile()}}
    # a real index_map (keys only need to pin the syntax): the star-tree test segment's
    lines = [l for l in open("/root/reference/pinot-segment-local/src/test/resources/data/startree/segment/index_map").read().splitlines()
             if l and not l.startswith("#")]
    out["_index_map_sample"] = {"_source": "pinot-segment-local/src/test/resources/data/startree/segment/index_map", "lines": lines[:12]}
    path = os.path.join(os.path.dirname(OUT), "pinot_v1_segment_directories.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


def export_raw_chunk_fixture():
    """pinot-segment-local/src/test/resources/data/fixedByteRaw.v2: a PASS_THROUGH version-2 fixed-byte chunk file written by
    the reference (2000 doubles, value i = i + 100.2356 -- FixedByteChunkSVForwardIndexTest.java:352-375
    testBackwardCompatibilityV2).  Pins the chunk header / offset table layout and the big-endian value bytes."""
    import base64
    src = "/root/reference/pinot-segment-local/src/test/resources/data/fixedByteRaw.v2"
    data = open(src, "rb").read()
    path = os.path.join(os.path.dirname(OUT), "fixedByteRaw_v2.json")
    with open(path, "w") as f:
        json.dump({"_source": src, "num_docs": 2000, "start_value": 100.2356, "stored_type": "DOUBLE",
                   "file_base64": base64.b64encode(data).decode()}, f)
    print("wrote", path, len(data), "bytes")


if __name__ == "__main__":
    main()
    export_prebuilt_segment()
    export_raw_chunk_fixture()
    export_segment_directories()
