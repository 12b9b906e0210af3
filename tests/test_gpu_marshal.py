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
