"""CPU tests of the JNI binding's native half (jni/): the flat-array marshalling of jni/pg_marshal.c, which the JNI functions of
jni/pinot_gpu_jni.c are thin wrappers around, and a type-check of those wrappers against the JNI stand-in header (no JDK exists here).
The GPU leg (tests/test_gpu_marshal.py) runs the reference's golden queries through the same layer."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import marshal as M
from pinot_amd import query as Q
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def same_query(a, b):
    assert (a.num_filter_nodes, a.num_predicates, a.num_aggregations, a.num_group_by, a.num_groups_limit, a.flags) == \
           (b.num_filter_nodes, b.num_predicates, b.num_aggregations, b.num_group_by, b.num_groups_limit, b.flags)
    for i in range(a.num_filter_nodes):
        assert (a.filter[i].op, a.filter[i].predicate, a.filter[i].num_children) == (b.filter[i].op, b.filter[i].predicate, b.filter[i].num_children)
    for i in range(a.num_predicates):
        pa, pb = a.predicates[i], b.predicates[i]
        assert (pa.kind, pa.column, pa.eval, pa.exclusive, pa.lo, pa.hi, pa.num_set_words) == (pb.kind, pb.column, pb.eval, pb.exclusive, pb.lo, pb.hi, pb.num_set_words)
        assert [pa.set_words[k] for k in range(pa.num_set_words)] == [pb.set_words[k] for k in range(pb.num_set_words)]
    for i in range(a.num_aggregations):
        assert (a.aggregations[i].function, a.aggregations[i].column) == (b.aggregations[i].function, b.aggregations[i].column)
    assert [a.group_by_columns[i] for i in range(a.num_group_by)] == [b.group_by_columns[i] for i in range(b.num_group_by)]


def golden_specs(seg):
    aggs = H.golden_aggregations(seg)
    c9 = seg.column_index("column9")
    return [Q.QuerySpec(aggs), Q.QuerySpec(aggs, filter=H.golden_filter_physical(seg)), Q.QuerySpec(aggs, filter=H.golden_filter(seg, True), group_by=[c9]),
            Q.QuerySpec([(Q.COUNT, -1)], filter=Q.not_(Q.leaf(H.in_pred(seg, "column6", [1689277, 2147419555], inverted=True))), num_groups_limit=7),
            Q.QuerySpec(aggs, filter=Q.leaf(Q.Pred.is_null(c9)), null_handling=True, group_by=[c9, seg.column_index("column11")])]


def test_query_arrays_become_the_same_pg_query():
    seg = H.golden_segment()
    for spec in golden_specs(seg):
        with M.MarshalledQuery(spec) as mq:
            same_query(mq.c, spec.c)


def test_inconsistent_arrays_are_refused():
    lib = M.load()
    z32, z64, zu = (C.c_int32 * 4)(), (C.c_int64 * 4)(), (C.c_uint32 * 4)()
    off = (C.c_int32 * 2)(0, 9)                                # one predicate whose set words would end past the array
    assert not lib.pgm_query_build(z32, 0, z32, z64, 1, off, zu, 4, z32, 0, z32, 0, 0, 0) and b"offsets" in lib.pgm_last_error()
    assert not lib.pgm_query_build(z32, -1, z32, z64, 0, off, zu, 0, z32, 0, z32, 0, 0, 0)
    assert not lib.pgm_query_build(None, 2, z32, z64, 0, off, zu, 0, z32, 0, z32, 0, 0, 0)


def test_segment_description_round_trip():
    lib = M.load()
    seg = H.golden_segment()
    n = len(seg.columns)
    names = (C.c_char_p * n)(*[c.name.encode() for c in seg.columns])
    ints = np.zeros(6 * n, dtype=np.int32)
    bufs = np.zeros(8 * n, dtype=np.int64)
    for i, c in enumerate(seg.columns):
        ints[6 * i:6 * i + 6] = (c.stored_type, c.encoding, c.bits, c.cardinality, 1, 0)
        bufs[8 * i:8 * i + 4] = (c.fwd.ctypes.data, c.fwd.nbytes, c.dictionary.ctypes.data, c.dictionary.nbytes)
        if c.inverted is not None:
            bufs[8 * i + 4:8 * i + 6] = (c.inverted.ctypes.data, c.inverted.nbytes)
    h = lib.pgm_segment_build(seg.name.encode(), 77, 0, seg.num_docs, n, names, ints.ctypes.data_as(C.POINTER(C.c_int32)), bufs.ctypes.data_as(C.POINTER(C.c_int64)))
    assert h
    d, want = lib.pgm_segment_get(h).contents, seg.desc
    assert (d.name, d.crc, d.num_docs, d.num_columns, d.device_id) == (seg.name.encode(), 77, seg.num_docs, n, 0)
    for i in range(n):
        a, b = d.columns[i], want.columns[i]
        assert (a.name, a.stored_type, a.fwd_encoding, a.bits_per_value, a.cardinality) == (b.name, b.stored_type, b.fwd_encoding, b.bits_per_value, b.cardinality)
        assert (a.fwd_data, a.fwd_size, a.dict_data, a.dict_size, a.inv_data, a.inv_size, a.null_data, a.null_size) == \
               (b.fwd_data, b.fwd_size, b.dict_data, b.dict_size, b.inv_data, b.inv_size, b.null_data, b.null_size)
    # and the description is usable as is: the oracle answers the golden query over it
    res = _abi.pg_result()
    spec = Q.QuerySpec(H.golden_aggregations(seg), filter=H.golden_filter_physical(seg))
    assert oracle.load().po_execute(lib.pgm_segment_get(h), C.byref(spec.c), C.byref(res)) == 0
    assert (res.stats.num_docs_scanned, res.stats.num_entries_scanned_in_filter) == (6129, 63064)
    oracle.load().po_result_free(C.byref(res))
    lib.pgm_segment_free(h)


def test_results_unpack_into_arrays():
    seg = H.golden_segment()
    g = H.load_golden_queries()
    for spec in golden_specs(seg)[:3]:
        res = _abi.pg_result()
        with M.MarshalledQuery(spec) as mq:
            assert oracle.load().po_execute(C.byref(seg.desc), C.byref(mq.c), C.byref(res)) == 0
        try:
            is_group_by = bool(spec.group_by)
            header, group_ids, counts, sums, sums_i64, sum_exact, mins, maxs = M.unpack_result(res, is_group_by)
            want = Q.Result(res, spec)
            assert tuple(header[:4]) == want.stats and header[M.H_FILTER_ENTRIES_EXACT] == 1 and header[M.H_IS_GROUP_BY] == int(is_group_by)
            na = len(spec.aggregations)
            rows = [(None, want.aggregations)] if not is_group_by else [(int(gid), want.groups[int(gid)]) for gid in group_ids]
            assert len(rows) == (1 if not is_group_by else len(want.groups))
            if is_group_by:
                keys = M.unpack_keys(res, len(spec.group_by))
                assert [tuple(int(x) for x in row) for row in keys] == want.group_keys and header[M.H_GROUP_KEY_KIND] == 0
            for r, (_, values) in enumerate(rows):
                for a, v in enumerate(values):
                    at = r * na + a
                    assert (counts[at], sums[at], sums_i64[at], bool(sum_exact[at]), mins[at], maxs[at]) == (v.count, v.sum, v.sum_i64, bool(v.sum_exact), v.min, v.max)
        finally:
            oracle.load().po_result_free(C.byref(res))
    # the unfiltered golden through the arrays: InnerSegmentAggregationSingleValueQueriesTest :44-61
    res = _abi.pg_result()
    spec = golden_specs(seg)[0]
    assert oracle.load().po_execute(C.byref(seg.desc), C.byref(spec.c), C.byref(res)) == 0
    header, _, counts, sums, _, _, mins, maxs = M.unpack_result(res, False)
    want = g["inner_segment"]["unfiltered"]
    assert (counts[0], sums[1], maxs[2], mins[3], sums[4], counts[4]) == (want["count"], float(want["sum_column1"]), float(want["max_column3"]),
                                                                      float(want["min_column6"]), float(want["avg_column7"][0]), want["avg_column7"][1])
    oracle.load().po_result_free(C.byref(res))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_jni_functions_type_check_against_the_stand_in_header():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "jni"), "-s", "check"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert out.returncode == 0, out.stdout.decode()
    # every native method PinotGpuNative.java declares has its JNI function, and the other way round
    java = open(os.path.join(ROOT, "java", "org", "apache", "pinot", "gpu", "PinotGpuNative.java")).read()
    c = open(os.path.join(ROOT, "jni", "pinot_gpu_jni.c")).read()
    import re
    declared = set(re.findall(r"static native [\w\[\]<>]+ (\w+)\(", java))
    defined = set(re.findall(r"Java_org_apache_pinot_gpu_PinotGpuNative_(\w+)\(", c))
    assert declared == defined and len(declared) >= 8, (declared, defined)


def test_keys_of_a_long_keyed_group_by_cross_as_dict_id_tuples():
    """A raw key beyond an int (LongMapBasedHolder): the int group ids are row numbers, the dictId tuples identify the groups."""
    import hash_holder_cases as HC
    seg, ids, specs = HC.build(HC.cases()[0])
    res = _abi.pg_result()
    spec = specs[2]                                                # numGroupsLimit 50
    with M.MarshalledQuery(spec) as mq:
        assert oracle.load().po_execute(C.byref(seg.desc), C.byref(mq.c), C.byref(res)) == 0
    try:
        header, group_ids, counts, *_ = M.unpack_result(res, True)
        keys = M.unpack_keys(res, len(spec.group_by))
        want = Q.Result(res, spec)
        assert header[M.H_GROUP_KEY_KIND] == 1 and header[M.H_NUM_GROUPS] == 50 and header[M.H_NUM_GROUPS_LIMIT_REACHED] == 1
        assert list(group_ids) == list(range(50))
        assert [tuple(int(x) for x in row) for row in keys] == want.group_keys
    finally:
        oracle.load().po_result_free(C.byref(res))
