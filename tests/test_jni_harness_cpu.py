"""CPU tests of the JNI functions' argument handling, EXECUTED through the JVM stand-in (jni/fake_jvm.c, pinot_amd/jni_harness.py):
what pinot_gpu_jni.c does before it reaches the device -- null arrays, lengths that are not whole records, mismatched batch arrays --
the exceptions it leaves pending, and that nothing stays pinned or referenced afterwards.  (The device half: tests/test_gpu_jni_harness.py.)"""
import ctypes as C

import numpy as np
import pytest

from pinot_amd import jni_harness as J
from pinot_amd import query as Q


@pytest.fixture(scope="module")
def jvm():
    return J.FakeJvm()


def _clean(jvm, refs_before):
    assert jvm.lib.fj_pins() == 0
    assert jvm.lib.fj_live_refs() == refs_before


def test_null_and_malformed_query_arrays_raise_the_documented_exceptions(jvm):
    before = jvm.lib.fj_live_refs()
    spec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(0, 1, 5)))
    arrays, limit, flags = jvm.query_arrays(spec)
    try:
        for missing in range(7):
            args = list(arrays)
            args[missing] = None
            with pytest.raises(J.JavaException) as e:
                jvm.call("queryCheck", C.c_int32, C.c_int64(0), *args, C.c_int32(limit), C.c_int32(flags))
            assert e.value.cls == "java/lang/NullPointerException"
            with pytest.raises(J.JavaException):
                jvm.call("execute", C.c_void_p, C.c_int64(0), *args, C.c_int32(limit), C.c_int32(flags))
        # lengths that are not whole records: filterNodes (3 ints each), predInts (4), aggregations (2)
        for slot, bad in ((0, [0, 0]), (1, [2, 0, 0]), (5, [0])):
            args = list(arrays)
            args[slot] = jvm.ints(bad)
            with pytest.raises(J.JavaException) as e:
                jvm.call("queryCheck", C.c_int32, C.c_int64(0), *args, C.c_int32(limit), C.c_int32(flags))
            assert e.value.cls == "java/lang/IllegalArgumentException"
            jvm.release(args[slot])
        # predicate arrays of different lengths
        args = list(arrays)
        args[2] = jvm.longs([1, 2, 3, 4])
        with pytest.raises(J.JavaException) as e:
            jvm.call("queryCheck", C.c_int32, C.c_int64(0), *args, C.c_int32(limit), C.c_int32(flags))
        assert e.value.cls == "java/lang/IllegalArgumentException"
        jvm.release(args[2])
        # set offsets that leave the word array (pg_marshal.c's own check, surfaced as IllegalArgumentException)
        args = list(arrays)
        args[3] = jvm.ints([0, 7])
        with pytest.raises(J.JavaException) as e:
            jvm.call("queryCheck", C.c_int32, C.c_int64(0), *args, C.c_int32(limit), C.c_int32(flags))
        assert e.value.cls == "java/lang/IllegalArgumentException" and "offsets" in e.value.message
        jvm.release(args[3])
    finally:
        jvm.release(*arrays)
    _clean(jvm, before)


def test_segment_open_refuses_null_and_mismatched_arrays(jvm):
    before = jvm.lib.fj_live_refs()
    name, names = jvm.string("seg"), jvm.objects([jvm.string("a"), jvm.string("b")])
    ints, bufs = jvm.ints(np.zeros(J.COLUMN_INTS * 2)), jvm.longs(np.zeros(J.COLUMN_BUFFERS * 2))
    short = jvm.ints(np.zeros(J.COLUMN_INTS))
    try:
        with pytest.raises(J.JavaException) as e:
            jvm.call("segmentOpen", C.c_int64, None, C.c_int64(0), C.c_int32(0), C.c_int32(10), names, ints, bufs)
        assert e.value.cls == "java/lang/NullPointerException"
        with pytest.raises(J.JavaException) as e:
            jvm.call("segmentOpen", C.c_int64, name, C.c_int64(0), C.c_int32(0), C.c_int32(10), names, short, bufs)
        assert e.value.cls == "java/lang/IllegalArgumentException"
    finally:
        jvm.release(name, names, ints, bufs, short)
    _clean(jvm, before)


def test_batch_refuses_malformed_batches_before_any_device_call(jvm):
    before = jvm.lib.fj_live_refs()
    spec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(0, 1, 5)))
    with pytest.raises(J.JavaException) as e:
        jvm.call("executeBatch", C.c_void_p, None, None)
    assert e.value.cls == "java/lang/NullPointerException"
    handles, queries = jvm.longs([0, 0, 0]), jvm.batch_queries([spec, spec])
    with pytest.raises(J.JavaException) as e:                                           # three handles, two queries
        jvm.call("executeBatch", C.c_void_p, handles, queries)
    assert e.value.cls == "java/lang/IllegalArgumentException"
    jvm.release(handles, queries)
    handles = jvm.longs([0, 0])
    arrays, limit, flags = jvm.query_arrays(spec)
    short_item = jvm.objects(arrays)                                                     # seven slots instead of PGM_QUERY_ARRAYS
    good_arrays, _, _ = jvm.query_arrays(spec)
    good_item = jvm.objects(good_arrays + [jvm.ints([limit, flags])])
    queries = jvm.objects([good_item, short_item])
    with pytest.raises(J.JavaException) as e:
        jvm.call("executeBatch", C.c_void_p, handles, queries)
    assert e.value.cls == "java/lang/IllegalArgumentException" and "PGM_QUERY_ARRAYS" in e.value.message
    jvm.release(queries)
    arrays, limit, flags = jvm.query_arrays(spec)
    bad_tail = jvm.objects(arrays + [jvm.ints([limit])])                                 # {numGroupsLimit} without flags
    good_arrays, _, _ = jvm.query_arrays(spec)
    queries = jvm.objects([jvm.objects(good_arrays + [jvm.ints([limit, flags])]), bad_tail])
    with pytest.raises(J.JavaException) as e:
        jvm.call("executeBatch", C.c_void_p, handles, queries)
    assert e.value.cls == "java/lang/IllegalArgumentException"
    jvm.release(handles, queries)
    # an empty batch is an empty answer
    handles, queries = jvm.longs([]), jvm.objects([])
    out = jvm.call("executeBatch", C.c_void_p, handles, queries)
    assert jvm.to_python(out) == []
    jvm.release(C.c_void_p(out), handles, queries)
    _clean(jvm, before)
    assert jvm.lib.fj_live_objects() == 0
