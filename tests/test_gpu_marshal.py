"""GPU leg of the JNI binding's native half: the reference's golden queries through jni/pg_marshal.c -- the arrays GpuQueryLowering.java
produces -> pgm_query_build -> pg_query_check / pg_execute -> pgm_result_fill -> the arrays GpuAggregationOperator.java reads, i.e. every
native step of PinotGpuNative.execute except the JNI array pinning itself."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import marshal as M
from pinot_amd import query as Q
import helpers as H

pytestmark = pytest.mark.gpu


def run(gseg, spec):
    res = _abi.pg_result()
    with M.MarshalledQuery(spec) as mq:
        assert gseg.lib.pg_query_check(gseg.handle, C.byref(mq.c)) == _abi.PG_OK
        _abi.check(gseg.lib, gseg.lib.pg_execute(gseg.handle, C.byref(mq.c), C.byref(res)))
    try:
        return M.unpack_result(res, bool(spec.group_by))
    finally:
        gseg.lib.pg_result_free(C.byref(res))


def test_reference_goldens_through_the_marshalling_layer(engine):
    g = H.load_golden_queries()
    seg = H.golden_segment()
    aggs = H.golden_aggregations(seg)
    with engine.open(seg) as gseg:
        for key, flt in (("unfiltered", None), ("filtered", H.golden_filter_physical(seg))):
            want = g["inner_segment"][key]
            header, _, counts, sums, sums_i64, sum_exact, mins, maxs = run(gseg, Q.QuerySpec(aggs, filter=flt))
            assert list(header[:4]) == want["stats"] and header[M.H_FILTER_ENTRIES_EXACT] == 1
            assert (counts[0], sums[1], sums_i64[1], maxs[2], mins[3], sums[4], counts[4]) == \
                   (want["count"], float(want["sum_column1"]), want["sum_column1"], float(want["max_column3"]), float(want["min_column6"]),
                    float(want["avg_column7"][0]), want["avg_column7"][1])
            # group-by column9: the golden row, and every row against the oracle
            gw = g["inner_segment_group_by_column9"][key]
            spec = Q.QuerySpec(aggs, filter=flt, group_by=[seg.column_index("column9")])
            header, group_ids, counts, sums, _, _, mins, maxs = run(gseg, spec)
            assert list(header[:4]) == gw["stats"] and header[M.H_IS_GROUP_BY] == 1 and header[M.H_NUM_GROUPS] == len(group_ids)
            gid = int(np.searchsorted(seg.column("column9").dict_values, gw["key"]))
            row = int(np.flatnonzero(group_ids == gid)[0])
            na = len(aggs)
            assert (counts[row * na], sums[row * na + 1], maxs[row * na + 2], mins[row * na + 3]) == \
                   (gw["count"], float(gw["sum_column1"]), float(gw["max_column3"]), float(gw["min_column6"]))
            ow = oracle.execute(seg, spec)
            assert sorted(int(x) for x in group_ids) == sorted(ow.groups)
            for r, gid in enumerate(group_ids):
                for a, v in enumerate(ow.groups[int(gid)]):
                    assert (counts[r * na + a], sums[r * na + a], mins[r * na + a], maxs[r * na + a]) == (v.count, v.sum, v.min, v.max)


def test_query_check_declines_what_execute_declines(engine):
    seg = H.golden_segment()
    c1 = seg.column_index("column1")
    nine = Q.and_(*[Q.leaf(Q.Pred.dict_range(c1, i, i + 100)) for i in range(9)])            # nine scan leaves: over the leaf table
    with engine.open(seg) as gseg:
        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=nine)
        res = _abi.pg_result()
        with M.MarshalledQuery(spec) as mq:
            assert gseg.lib.pg_query_check(gseg.handle, C.byref(mq.c)) == _abi.PG_ERR_UNSUPPORTED
            assert gseg.lib.pg_execute(gseg.handle, C.byref(mq.c), C.byref(res)) == _abi.PG_ERR_UNSUPPORTED
        eight = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(*nine.children[:8]))
        assert gseg.check(eight) == _abi.PG_OK and gseg.execute(eight).stats[0] == oracle.execute(seg, eight).stats[0]


def test_a_batch_through_the_marshalling_layer(engine):
    """The native steps of PinotGpuNative.executeBatch (jni/pinot_gpu_jni.c) except the JNI array handling: one pgm_query_build per item,
    ONE pg_execute_batch over the built queries, pgm_result_* per item -- equal to one pg_execute per item, statistics included; an item
    the device declines fails alone."""
    seg = H.golden_segment()
    aggs = H.golden_aggregations(seg)
    c1, c9 = seg.column_index("column1"), seg.column_index("column9")
    nine = Q.and_(*[Q.leaf(Q.Pred.dict_range(c1, i, i + 100)) for i in range(9)])            # over the leaf table: PG_ERR_UNSUPPORTED
    specs = [Q.QuerySpec(aggs), Q.QuerySpec(aggs, filter=H.golden_filter_physical(seg)), Q.QuerySpec([(Q.COUNT, -1)], filter=nine),
             Q.QuerySpec([(Q.SUM, c1), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(c1, 10, 4000))),
             Q.QuerySpec(aggs, group_by=[c9])]
    with engine.open(seg) as gseg:
        built = [M.MarshalledQuery(s) for s in specs]
        try:
            n = len(specs)
            handles = (C.c_void_p * n)(*[gseg.handle] * n)
            queries = (C.POINTER(_abi.pg_query) * n)(*[C.pointer(b.c) for b in built])
            results = (_abi.pg_result * n)()
            statuses = (C.c_int * n)()
            assert gseg.lib.pg_execute_batch(handles, queries, n, results, statuses) == _abi.PG_OK
            assert list(statuses) == [_abi.PG_OK, _abi.PG_OK, _abi.PG_ERR_UNSUPPORTED, _abi.PG_OK, _abi.PG_OK]
            assert b"batch item 2" in gseg.lib.pg_last_error()
            for i, spec in enumerate(specs):
                if statuses[i] != _abi.PG_OK:
                    continue
                got = M.unpack_result(results[i], bool(spec.group_by))
                keys = M.unpack_keys(results[i], len(spec.group_by)) if spec.group_by else None
                gseg.lib.pg_result_free(C.byref(results[i]))
                want = run(gseg, spec)
                got[0][M.H_DOMINANT_KERNEL] = want[0][M.H_DOMINANT_KERNEL] = 0      # (a diagnostic: the shared launch is a kernel of its own)
                for a, b in zip(got, want):
                    assert np.array_equal(a, b), i
                if keys is not None:
                    assert keys.shape == (len(got[1]), 1) and np.array_equal(keys[:, 0], got[1])      # one int key column: the raw key is the dictId
        finally:
            for b in built:
                b.close()
