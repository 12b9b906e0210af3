"""LONG / FLOAT / DOUBLE stored types in the oracle (CPU): layouts against the reference's own files, typed dictionary
search, aggregation semantics against plain numpy, and the product's C++ writers against the oracle's."""
import base64
import json
import os

import numpy as np
import pytest

import helpers as H
from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S


def _fixture(name):
    return json.load(open(os.path.join(H.GOLDEN_DIR, name)))


def test_raw_chunk_file_written_by_the_reference():
    """fixedByteRaw.v2 (FixedByteChunkSVForwardIndexTest.java:352-375): header fields, chunk offset table and the
    big-endian doubles i + 100.2356 -- pins BaseChunkForwardIndexReader's layout byte for byte."""
    g = _fixture("fixedByteRaw_v2.json")
    data = np.frombuffer(base64.b64decode(g["file_base64"]), dtype=np.uint8).copy()
    version, num_chunks, docs_per_chunk, entry, total, compression, header_start, raw_start = oracle.raw_header(data)
    assert (version, num_chunks, docs_per_chunk, entry, total, compression, header_start) == (2, 2, 1000, 8, 2000, 0, 28)
    assert raw_start == 28 + 4 * num_chunks and data.nbytes == raw_start + 8 * 2000
    # the chunk offset table: 4-byte big-endian file positions (version 2)
    offs = data[28:28 + 8].view(">i4").tolist()
    assert offs == [raw_start, raw_start + 8 * 1000]
    col = S.Column("d", _abi.PG_FWD_RAW_FIXED_BYTE, 64, 0, data, stored_type=_abi.PG_TYPE_DOUBLE)
    seg = S.SegmentData("fixedByteRaw", 2000, [col])
    got, _ = oracle.read_double_values(seg, 0, np.arange(2000, dtype=np.int32))
    want = np.arange(2000, dtype=np.float64) + g["start_value"]
    assert np.array_equal(got, want)
    # the writers (oracle restatement and product) reproduce the reference's file
    assert np.array_equal(oracle.raw_write_typed(want, 1000), data)
    assert np.array_equal(S.Column.raw_typed("d", want, 1000).fwd, data)
    r = oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0), (Q.MIN, 0), (Q.MAX, 0)]))
    assert r.intermediates()[0] == 2000 and r.intermediates()[2:] == [100.2356, 2099.2356]
    assert r.intermediates()[1] == float(np.add.accumulate(want)[-1])     # the doc-order double sum


def test_long_and_float_dictionaries_written_by_the_reference():
    """paddingOld's LONG (`outgoingName1`) and FLOAT (`percent`) columns: LongDictionary / FloatDictionary bytes."""
    g = _fixture("pinot_v1_segment_paddingOld.json")
    n = g["total_docs"]
    cols = []
    for name, st, dt in (("outgoingName1", _abi.PG_TYPE_LONG, ">i8"), ("percent", _abi.PG_TYPE_FLOAT, ">f4")):
        c = g["columns"][name]
        assert c["dataType"] == ("LONG" if st == _abi.PG_TYPE_LONG else "FLOAT")
        dict_bytes = np.frombuffer(bytes.fromhex(c["dict_hex"]), dtype=np.uint8).copy()
        values = dict_bytes.view(dt)
        assert np.all(np.diff(values.astype(np.float64)) > 0) and values.shape[0] == c["cardinality"]
        assert np.array_equal(oracle.dict_write_typed(values.astype(dt[1:])), dict_bytes)
        cols.append(S.Column(name, _abi.PG_FWD_FIXED_BIT_DICT, c["bitsPerElement"], c["cardinality"],
                             np.frombuffer(bytes.fromhex(c["fwd_hex"]), dtype=np.uint8).copy(), dict_bytes, stored_type=st))
    seg = S.SegmentData("paddingOld", n, cols)
    docs = np.arange(n, dtype=np.int32)
    longs = cols[0].dictionary.view(">i8").astype(np.int64)
    ids0 = oracle.read_dict_ids(cols[0].fwd, cols[0].bits, n, docs)
    ids1 = oracle.read_dict_ids(cols[1].fwd, cols[1].bits, n, docs)
    assert sorted(ids0.tolist()) == list(range(5)) and sorted(ids1.tolist()) == list(range(5))
    dv, lv = oracle.read_double_values(seg, 0, docs)
    assert lv.tolist() == longs[ids0].tolist() and dv.tolist() == [float(x) for x in longs[ids0]]
    floats = cols[1].dictionary.view(">f4").astype(np.float32)
    dv, _ = oracle.read_double_values(seg, 1, docs)
    assert dv.tolist() == [float(x) for x in floats[ids1]]
    r = oracle.execute(seg, Q.QuerySpec([(Q.SUM, 0), (Q.MAX, 0), (Q.SUM, 1), (Q.MIN, 1)]))
    assert r.intermediates()[0] == float(longs.sum()) and r.intermediates()[1] == float(longs.max())
    assert r.intermediates()[2] == float(np.add.accumulate(floats[ids1].astype(np.float64))[-1]) and r.intermediates()[3] == float(floats.min())


TYPED = [(np.int64, "LONG narrow"), (np.int64, "LONG wide"), (np.float32, "FLOAT"), (np.float64, "DOUBLE")]


def typed_values(rng, dtype, label, n, card=200):
    if dtype == np.int64 and "wide" in label:
        pool = rng.integers(-2 ** 40, 2 ** 40, card)      # range >= 2^31: no offset dictionary; sums stay inside int64
    elif dtype == np.int64:
        pool = 1_600_000_000_000 + rng.integers(0, 2 ** 30, card)
    elif dtype == np.float32:
        pool = (rng.random(card) * 1000 - 100).astype(np.float32)
    else:
        pool = rng.random(card) * 1e6 - 1e5
    pool = np.unique(pool.astype(dtype))
    return pool[rng.integers(0, pool.shape[0], n)].astype(dtype)


def doc_order_sum(values, block=10000):
    """SumAggregationFunction: a double innerSum per 10 000-doc block, added to the holder (sum + holder)."""
    v = values.astype(np.float64)
    holder = 0.0
    for s in range(0, v.shape[0], block):
        inner = 0.0
        for x in v[s:s + block].tolist():
            inner += x
        holder = inner + holder
    return holder


@pytest.mark.parametrize("dtype,label", TYPED)
@pytest.mark.parametrize("raw", [False, True])
def test_typed_aggregations_match_numpy(dtype, label, raw):
    rng = np.random.default_rng(hash(label) % 1000 + int(raw))
    n = 25_000
    values = typed_values(rng, dtype, label, n)
    fvals = rng.integers(0, 50, n).astype(np.int32)
    col = S.Column.raw_typed("m", values) if raw else S.Column.dict_encoded_typed("m", values)
    seg = S.SegmentData("t", n, [col, S.Column.dict_encoded("f", fvals)])
    sel = fvals < 20
    s, e = oracle.lower_range(seg.columns[1].dictionary, seg.columns[1].cardinality, None, 20, True, False)
    spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0), (Q.MIN, 0), (Q.MAX, 0), (Q.AVG, 0)], filter=Q.leaf(Q.Pred.dict_range(1, s, e)))
    r = oracle.execute(seg, spec)
    m = values[sel]
    assert r.aggregations[0].count == int(sel.sum())
    assert r.aggregations[1].sum == doc_order_sum(m)
    assert r.aggregations[2].min == float(m.min()) and r.aggregations[3].max == float(m.max())
    assert r.aggregations[4].count == int(sel.sum())
    if dtype == np.int64:
        assert r.aggregations[1].sum_exact and r.aggregations[1].sum_i64 == int(m.astype(object).sum()) % 2 ** 64 - (2 ** 64 if int(m.astype(object).sum()) % 2 ** 64 >= 2 ** 63 else 0)
    else:
        assert not r.aggregations[1].sum_exact
    # group by the filter column
    rg = oracle.execute(seg, Q.QuerySpec([(Q.SUM, 0), (Q.MAX, 0)], group_by=[1])) if not raw else None
    if rg is not None:
        for gid, vals in rg.groups.items():
            mm = values[fvals == seg.columns[1].dict_values[gid]]
            assert vals[1].max == float(mm.max())
            assert np.isclose(vals[0].sum, float(mm.astype(np.float64).sum()), rtol=1e-12)


@pytest.mark.parametrize("dtype,label", TYPED)
def test_typed_range_lowering(dtype, label):
    rng = np.random.default_rng(5)
    values = typed_values(rng, dtype, label, 5000)
    col = S.Column.dict_encoded_typed("m", values)
    d = col.dict_values
    for lo_i, hi_i in ((0, len(d) - 1), (3, 17), (10, 10), (len(d) // 2, len(d) - 2)):
        for li in (True, False):
            for ui in (True, False):
                lo, hi = d[lo_i], d[hi_i]
                s, e = oracle.lower_range_typed(col, lo, li, hi, ui)
                want = np.nonzero((d >= lo if li else d > lo) & (d <= hi if ui else d < hi))[0]
                assert (s, max(e, s)) == ((int(want[0]), int(want[-1]) + 1) if want.size else (s, s)), (lo_i, hi_i, li, ui)
    # a bound that is not in the dictionary -> insertion point
    if dtype != np.int64:
        mid = (float(d[4]) + float(d[5])) / 2
        if dtype == np.float32:
            mid = float(np.float32(mid))
        if d[4] < mid < d[5]:
            assert oracle.lower_range_typed(col, mid, True, None, True) == (5, len(d))
            assert oracle.lower_range_typed(col, None, True, mid, True) == (0, 5)


def test_typed_writers_product_equals_oracle():
    rng = np.random.default_rng(9)
    for dtype, label in TYPED:
        values = typed_values(rng, dtype, label, 3333)
        assert np.array_equal(S.Column.raw_typed("r", values, 1000).fwd, oracle.raw_write_typed(values, 1000))
        col = S.Column.dict_encoded_typed("d", values)
        assert np.array_equal(col.dictionary, oracle.dict_write_typed(col.dict_values))


def test_raw_long_and_floating_point_range_predicates():
    rng = np.random.default_rng(11)
    values = rng.integers(-2 ** 40, 2 ** 40, 20_000).astype(np.int64)
    dv = values.astype(np.float64) / 7
    dv[::97] = np.nan
    dv[1::97] = -0.0
    dv[2::97] = 0.0
    fv = (values % 1000).astype(np.float32) / 8
    seg = S.SegmentData("t", values.shape[0], [S.Column.raw_typed("l", values), S.Column.raw_typed("d", dv), S.Column.raw_typed("f", fv)])
    lo, hi = -2 ** 39, 2 ** 38
    r = oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], filter=Q.leaf(Q.Pred.raw_range(0, lo, hi))))
    sel = (values >= lo) & (values <= hi)
    assert r.aggregations[0].count == int(sel.sum()) and r.aggregations[1].sum_i64 == int(values[sel].sum())
    # DoubleRawValueBasedRangePredicateEvaluator: primitive compares (NaN never matches, -0.0 == 0.0)
    for dlo, dhi in ((-1e9, 1e9), (0.0, 5e10), (-np.inf, -0.0), (-np.inf, np.inf), (3.0, 2.0)):
        r = oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.raw_range_f64(1, dlo, dhi))))
        with np.errstate(invalid="ignore"):
            assert r.aggregations[0].count == int(((dv >= dlo) & (dv <= dhi)).sum()), (dlo, dhi)
    r = oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.raw_range_f64(2, 10.125, 100.0))))
    assert r.aggregations[0].count == int(((fv >= np.float32(10.125)) & (fv <= np.float32(100.0))).sum())


def test_doc_range_predicate_of_sorted_columns():
    """PG_PRED_DOC_RANGE: the docId range SortedIndexBasedFilterOperator derives from a sorted column's [start, end] pairs."""
    rng = np.random.default_rng(17)
    n = 30_000
    v = rng.integers(0, 1000, n).astype(np.int32)
    seg = S.SegmentData("sorted", n, [S.Column.dict_encoded("v", v)])
    docs = np.arange(n)
    for lo, hi, excl in ((0, n - 1, False), (100, 100, False), (2047, 2049, False), (5000, 20_000, False), (5000, 20_000, True), (0, 0, True),
                         (n - 5, n + 100, False), (-7, 3, False), (10, 9, False)):
        r = oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], filter=Q.leaf(Q.Pred.doc_range(lo, hi, exclusive=excl))))
        sel = (docs >= lo) & (docs <= hi)
        if excl:
            sel = ~sel
        assert r.aggregations[0].count == int(sel.sum()) and r.aggregations[1].sum_i64 == int(v[sel].astype(np.int64).sum())
        assert r.stats[1] == 0      # numEntriesScannedInFilter: a sorted index scans nothing
