"""The native segment-directory loader on the host side only: metadata, layouts, dictionaries."""
import numpy as np
import pytest

import segment_dirs as D
from pinot_amd import host
from pinot_amd import segment as S


@pytest.mark.parametrize("name,names", [("paddingOld", ["lynda 2.0", "lynda"]), ("paddingPercent", ["lynda 2.0", "lynda"]),
                                        ("paddingNull", ["lynda", "lynda 2.0"])])
def test_reference_v1_directories(tmp_path, name, names):
    """The three 5-doc segments the reference's Java creator wrote (INT / FLOAT / LONG / STRING columns; '%' and NUL padding)."""
    seg = host.DirectorySegment(D.write_reference_directory(tmp_path, name), device=-1)
    try:
        d = seg.describe()
        assert d["totalDocs"] == 5 and d["notOffloaded"] == []
        cols = {c["name"]: c for c in d["columns"]}
        assert {n: c["dataType"] for n, c in cols.items()} == {"age": "INT", "name": "STRING", "outgoingName1": "LONG", "percent": "FLOAT"}
        assert all(c["hasDictionary"] and not c["hasInvertedIndex"] for c in cols.values())      # metadata claims inverted indexes, no files
        assert (cols["age"]["cardinality"], cols["age"]["bitsPerElement"]) == (5, 3) and cols["name"]["cardinality"] == 2
        assert [cols["name"]["minValue"], cols["name"]["maxValue"]] == names
        assert (cols["outgoingName1"]["minValue"], cols["outgoingName1"]["maxValue"]) == ("246", "902")
    finally:
        seg.destroy()


def _synthetic_columns():
    rng = np.random.default_rng(3)
    n = 5000
    k = np.sort(rng.integers(0, 40, n)).astype(np.int32) * 5 + 1
    return n, k, [S.Column.dict_encoded("k", k), S.Column.dict_encoded("v", rng.integers(0, 10_000, n).astype(np.int32), with_inverted=True),
                  S.Column.dict_encoded_typed("d", rng.random(n).round(3)), S.Column.raw_typed("r", rng.integers(-2 ** 40, 2 ** 40, n).astype(np.int64))]


def test_v3_container_and_sorted_forward_index(tmp_path):
    n, k, cols = _synthetic_columns()
    # the index_map syntax is the reference's (sample lines of a real index_map in the fixture)
    sample = D.reference_directories()["_index_map_sample"]["lines"]
    assert sample[0].endswith(".dictionary.startOffset = 0") and " = " in sample[1] and sample[1].split(" = ")[0].endswith(".dictionary.size")
    v3 = host.DirectorySegment(D.write_v3(tmp_path, "seg_v3", n, cols), device=-1)
    kcol = cols[0]
    ids = np.searchsorted(kcol.dict_values, k).astype(np.int32)
    v1 = host.DirectorySegment(D.write_v1(tmp_path, "seg_v1_sorted", n, cols, sorted_fwd={"k": D.sorted_forward_index(ids, kcol.cardinality)}), device=-1)
    try:
        for seg in (v3, v1):
            d = seg.describe()
            assert d["totalDocs"] == n and d["notOffloaded"] == []
            c = {x["name"]: x for x in d["columns"]}
            assert [c[x]["dataType"] for x in ("k", "v", "d", "r")] == ["INT", "INT", "DOUBLE", "LONG"]
            assert c["v"]["hasInvertedIndex"] and not c["r"]["hasDictionary"] and c["k"]["bitsPerElement"] == kcol.bits
    finally:
        v3.destroy()
        v1.destroy()


def test_loader_errors(tmp_path):
    with pytest.raises(host.HostError):
        host.DirectorySegment(str(tmp_path / "missing"), device=-1)
    n, _, cols = _synthetic_columns()
    root = D.write_v3(tmp_path, "corrupt", n, cols)
    with open(root + "/v3/columns.psf", "r+b") as f:      # break the first magic marker
        f.write(b"\0")
    with pytest.raises(host.HostError) as ei:
        host.DirectorySegment(root, device=-1)
    assert "magic marker" in str(ei.value)
