"""CPU tests of the DataTable V4 writer (pinot_amd/csrc/host/datatable_v4.cpp, C entry ph_datatable_v4_build): byte for byte against the
independent Python restatement in tests/datatable_v4.py, and decoded back with the reader of the same file.  The reference holds no
serialized DataTable fixtures and no JVM exists here: parity of the BYTES is unpinned (two restatements agreeing), which the fixture
tests/golden/datatable_v4_golden.json records for the reference's golden query."""
import ctypes as C
import json
import os

import numpy as np

from pinot_amd import segment as S
import datatable_v4 as D
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEY_TYPE = {D.INT: 0, D.LONG: 1, D.FLOAT: 2, D.DOUBLE: 3, D.STRING: 4}


def build(functions, columns, keys, key_types, rows, stats, null_handling=False, limit_reached=False, processed=1, matched=1):
    """rows: [(key values..., [(count, sum, min, max, is_null) per function])].  Returns the C++ writer's bytes."""
    lib = S.load_host_library()
    lib.ph_datatable_v4_build.restype = C.POINTER(C.c_uint8)
    P = C.POINTER
    lib.ph_datatable_v4_build.argtypes = [C.c_int32, C.c_int32, P(C.c_int32), P(C.c_char_p), C.c_int32, P(C.c_char_p), P(C.c_int32), C.c_int64, P(C.c_int64), P(C.c_double),
                                          P(C.c_char_p), P(C.c_int64), P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_uint8), P(C.c_int64), C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, P(C.c_int64), P(C.c_int32)]
    nf, nk, nr = len(functions), len(keys), len(rows)
    fn = (C.c_int32 * max(nf, 1))(*functions)
    cols = (C.c_char_p * max(nf, 1))(*[c.encode() for c in columns])
    kn = (C.c_char_p * max(nk, 1))(*[k.encode() for k in keys])
    kt = (C.c_int32 * max(nk, 1))(*[KEY_TYPE[t] for t in key_types])
    kl, kd = (C.c_int64 * max(nr * nk, 1))(), (C.c_double * max(nr * nk, 1))()
    ks = (C.c_char_p * max(nr * nk, 1))()
    counts, sums = (C.c_int64 * max(nr * nf, 1))(), (C.c_double * max(nr * nf, 1))()
    mins, maxs = (C.c_double * max(nr * nf, 1))(), (C.c_double * max(nr * nf, 1))()
    nulls = (C.c_uint8 * max(nr * nf, 1))()
    for r, (key_values, cells) in enumerate(rows):
        for k, v in enumerate(key_values):
            if key_types[k] in (D.INT, D.LONG):
                kl[r * nk + k] = int(v)
            elif key_types[k] == D.STRING:
                ks[r * nk + k] = v.encode()
            else:
                kd[r * nk + k] = float(v)
        for f, (c, s, mn, mx, is_null) in enumerate(cells):
            counts[r * nf + f], sums[r * nf + f], mins[r * nf + f], maxs[r * nf + f], nulls[r * nf + f] = c, s, mn, mx, int(is_null)
    st = (C.c_int64 * 4)(*stats)
    size, status = C.c_int64(), C.c_int32()
    ptr = lib.ph_datatable_v4_build(int(bool(keys)), nf, fn, cols, nk, kn, kt, nr, kl, kd, ks, counts, sums, mins, maxs, nulls, st, int(null_handling), int(limit_reached),
                                    processed, matched, C.byref(size), C.byref(status))
    assert status.value == 0 and ptr
    data = bytes(bytearray(ptr[:size.value]))
    lib.ph_free.argtypes = [C.c_void_p]
    lib.ph_free(C.cast(ptr, C.c_void_p))
    return data


def python_rows(functions, rows):
    out = []
    for key_values, cells in rows:
        row = list(key_values)
        for f, (c, s, mn, mx, is_null) in zip(functions, cells):
            if is_null:
                row.append(None)
            else:
                row.append({D.AGG_COUNT: c, D.AGG_SUM: s, D.AGG_MIN: mn, D.AGG_MAX: mx}.get(f, (s, c)))
        out.append(row)
    return out


def schema(functions, columns, keys, key_types):
    return list(keys) + ["%s(%s)" % (D.AGG_NAME[f], c) for f, c in zip(functions, columns)], list(key_types) + [D.INTERMEDIATE[f] for f in functions]


def test_aggregation_block_of_the_reference_golden_query():
    g = H.load_golden_queries()["inner_segment"]["filtered"]
    functions, columns = [D.AGG_COUNT, D.AGG_SUM, D.AGG_MAX, D.AGG_MIN, D.AGG_AVG], ["*", "column1", "column3", "column6", "column7"]
    cells = [(g["count"], 0.0, 0.0, 0.0, False), (0, float(g["sum_column1"]), 0.0, 0.0, False), (0, 0.0, 0.0, float(g["max_column3"]), False),
             (0, 0.0, float(g["min_column6"]), 0.0, False), (g["avg_column7"][1], float(g["avg_column7"][0]), 0.0, 0.0, False)]
    rows = [((), cells)]
    data = build(functions, columns, [], [], rows, g["stats"])
    names, types = schema(functions, columns, [], [])
    want = D.encode(names, types, python_rows(functions, rows), D.results_metadata(g["stats"], 1, 1), group_by=False)
    assert data == want
    back = D.decode(data)
    assert back["names"] == ["count(*)", "sum(column1)", "max(column3)", "min(column6)", "avg(column7)"] and back["types"] == [D.LONG, D.DOUBLE, D.DOUBLE, D.DOUBLE, D.OBJECT]
    assert back["rows"] == [[6129, 6875947596072.0, float(g["max_column3"]), float(g["min_column6"]), (float(g["avg_column7"][0]), g["avg_column7"][1])]]
    assert back["metadata"] == {"totalDocs": 30000, "numDocsScanned": 6129, "numEntriesScannedInFilter": 63064, "numEntriesScannedPostFilter": 24516,
                                "numSegmentsProcessed": 1, "numSegmentsMatched": 1, "numConsumingSegmentsProcessed": 0, "numConsumingSegmentsMatched": 0}
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "datatable_v4_golden.json")))
    assert data.hex() == fixture["inner_segment_filtered_aggregation"]["hex"]


def test_group_by_blocks_with_every_key_type_and_null_results():
    rng = np.random.default_rng(4)
    functions, columns = [D.AGG_SUM, D.AGG_AVG, D.AGG_COUNT, D.AGG_MIN, D.AGG_MAX], ["a", "b", "*", "c", "c"]
    keys, key_types = ["ki", "ks", "kl", "kd", "kf"], [D.INT, D.STRING, D.LONG, D.DOUBLE, D.FLOAT]
    words = ["", "P", "gFuH", "o", "t", "café", "P"]
    rows = []
    for r in range(57):
        key_values = (int(rng.integers(-2**31, 2**31)), words[r % len(words)], int(rng.integers(-2**62, 2**62)), float(rng.normal()) * 1e9, float(np.float32(rng.normal())))
        cells = [(int(rng.integers(0, 10**9)), float(rng.normal()) * 1e12, float(rng.normal()), float(rng.normal()), bool(r % 11 == 3 and f != 2)) for f in range(5)]
        rows.append((key_values, cells))
    stats = [123456, 7, 617280, 10**9]
    for null_handling in (False, True):
        use = rows if null_handling else [(k, [(c, s, mn, mx, False) for (c, s, mn, mx, _) in cells]) for k, cells in rows]
        data = build(functions, columns, keys, key_types, use, stats, null_handling=null_handling, limit_reached=null_handling, processed=8, matched=5)
        names, types = schema(functions, columns, keys, key_types)
        want = D.encode(names, types, python_rows(functions, use), D.results_metadata(stats, 8, 5, group_by=True, limit_reached=null_handling), null_handling=null_handling)
        assert data == want
        back = D.decode(data)
        assert back["types"] == [D.INT, D.STRING, D.LONG, D.DOUBLE, D.FLOAT, D.DOUBLE, D.OBJECT, D.LONG, D.DOUBLE, D.DOUBLE]
        assert len(back["rows"]) == 57 and back["rows"][5][1] == "café" and back["rows"][0][0] == rows[0][0][0]
        assert back["metadata"].get("numGroupsLimitReached") == ("true" if null_handling else None) and back["metadata"]["numResizes"] == 0
        if null_handling:
            nulls = [r for r in range(57) if r % 11 == 3]
            assert back["null_rows"][5] == nulls and back["null_rows"][7] == [] and back["null_rows"][6] == []      # OBJECT nulls travel in the object itself
            assert back["rows"][3][6] is None and back["rows"][3][5] == 0.0
        else:
            assert back["null_rows"] is None


def test_aggregation_block_with_null_results_under_null_handling():
    """AggregationResultsBlock.getDataTable (:113-131): every null result, a null AvgPair included, is row 0 of its column's null bitmap."""
    functions, columns = [D.AGG_COUNT, D.AGG_SUM, D.AGG_AVG, D.AGG_MIN, D.AGG_AVG], ["*", "a", "a", "b", "b"]
    cells = [(0, 0.0, 0.0, 0.0, False), (0, 0.0, 0.0, 0.0, True), (0, 0.0, 0.0, 0.0, True), (0, 0.0, 2.5, 0.0, False), (3, 7.5, 0.0, 0.0, False)]
    rows = [((), cells)]
    stats = [0, 10, 0, 10]
    data = build(functions, columns, [], [], rows, stats, null_handling=True)
    names, types = schema(functions, columns, [], [])
    assert data == D.encode(names, types, python_rows(functions, rows), D.results_metadata(stats, 1, 1), null_handling=True, group_by=False)
    back = D.decode(data)
    assert back["null_rows"] == [[], [0], [0], [], []]
    assert back["rows"] == [[0, 0.0, None, 2.5, (7.5, 3)]]


def test_empty_group_by_block():
    data = build([D.AGG_COUNT], ["*"], ["k"], [D.INT], [], [0, 0, 0, 5], processed=1, matched=0)
    back = D.decode(data)
    assert back["rows"] == [] and back["names"] == ["k", "count(*)"] and back["metadata"]["numSegmentsMatched"] == 0
    assert data == D.encode(["k", "count(*)"], [D.INT, D.LONG], [], D.results_metadata([0, 0, 0, 5], 1, 0, group_by=True))
