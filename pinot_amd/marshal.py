"""ctypes handle over jni/libpinot_gpu_marshal.so (jni/pg_marshal.c): the flat-array marshalling the JNI shim calls, driven from Python so that
it is exercised where no JDK exists.  `flatten(spec)` produces the arrays java/org/apache/pinot/gpu/GpuQueryLowering.java produces."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi

JNI_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jni")
LIB_PATH = os.path.join(JNI_DIR, "libpinot_gpu_marshal.so")
HEADER_LEN = 13
(H_NUM_DOCS_SCANNED, H_ENTRIES_IN_FILTER, H_ENTRIES_POST_FILTER, H_TOTAL_DOCS, H_FILTER_ENTRIES_EXACT, H_NUM_AGGREGATIONS, H_NUM_GROUPS,
 H_GROUP_ID_UPPER_BOUND, H_NUM_GROUPS_LIMIT_REACHED, H_DOMINANT_KERNEL, H_IS_GROUP_BY, H_GROUP_KEY_KIND, H_NUM_GROUP_BY) = range(HEADER_LEN)

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", JNI_DIR, "-s", "marshal"])
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.pgm_query_build.restype = C.c_void_p
    lib.pgm_query_build.argtypes = [P(C.c_int32), C.c_int32, P(C.c_int32), P(C.c_int64), C.c_int32, P(C.c_int32), P(C.c_uint32), C.c_int32,
                                    P(C.c_int32), C.c_int32, P(C.c_int32), C.c_int32, C.c_int32, C.c_int32]
    lib.pgm_query_get.restype = P(_abi.pg_query)
    lib.pgm_query_get.argtypes = [C.c_void_p]
    lib.pgm_query_free.argtypes = [C.c_void_p]
    lib.pgm_segment_build.restype = C.c_void_p
    lib.pgm_segment_build.argtypes = [C.c_char_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, P(C.c_char_p), P(C.c_int32), P(C.c_int64)]
    lib.pgm_segment_get.restype = P(_abi.pg_segment_desc)
    lib.pgm_segment_get.argtypes = [C.c_void_p]
    lib.pgm_segment_free.argtypes = [C.c_void_p]
    lib.pgm_result_rows.restype = C.c_int64
    lib.pgm_result_rows.argtypes = [P(_abi.pg_result), C.c_int32]
    lib.pgm_result_header.argtypes = [P(_abi.pg_result), C.c_int32, P(C.c_int64)]
    lib.pgm_result_fill.restype = C.c_int64
    lib.pgm_result_fill.argtypes = [P(_abi.pg_result), C.c_int32, P(C.c_int32), P(C.c_int64), P(C.c_double), P(C.c_int64), P(C.c_int32), P(C.c_double), P(C.c_double)]
    lib.pgm_result_fill_keys.restype = C.c_int64
    lib.pgm_result_fill_keys.argtypes = [P(_abi.pg_result), C.c_int32, P(C.c_int32)]
    lib.pgm_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def flatten(spec):
    """A QuerySpec as the arrays of jni/pg_marshal.h."""
    q = spec.c
    nodes = np.zeros(3 * q.num_filter_nodes, dtype=np.int32)
    for i in range(q.num_filter_nodes):
        nodes[3 * i:3 * i + 3] = (q.filter[i].op, q.filter[i].predicate, q.filter[i].num_children)
    pred_ints = np.zeros(4 * q.num_predicates, dtype=np.int32)
    pred_longs = np.zeros(2 * q.num_predicates, dtype=np.int64)
    offsets = np.zeros(q.num_predicates + 1, dtype=np.int32)
    words = []
    for i, p in enumerate(spec.predicates):
        cp = q.predicates[i]
        pred_ints[4 * i:4 * i + 4] = (cp.kind, cp.column, cp.eval, cp.exclusive)
        pred_longs[2 * i:2 * i + 2] = (cp.lo, cp.hi)
        if p.set_words is not None:
            words.append(np.asarray(p.set_words, dtype=np.uint32))
        offsets[i + 1] = offsets[i] + (0 if p.set_words is None else int(p.set_words.shape[0]))
    set_words = np.concatenate(words) if words else np.zeros(0, dtype=np.uint32)
    aggs = np.array([x for f, c in spec.aggregations for x in (f, c)], dtype=np.int32)
    group_by = np.array(spec.group_by, dtype=np.int32)
    return dict(nodes=nodes, pred_ints=pred_ints, pred_longs=pred_longs, set_offsets=offsets, set_words=set_words, aggregations=aggs,
                group_by=group_by, num_groups_limit=int(q.num_groups_limit), flags=int(q.flags))


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class MarshalledQuery:
    """pgm_query_build over flatten(spec); .c is the pg_query the marshalling layer built (valid until close())."""

    def __init__(self, spec):
        self.lib = load()
        self.arrays = f = flatten(spec)
        self.handle = self.lib.pgm_query_build(_p(f["nodes"], C.c_int32), len(f["nodes"]) // 3, _p(f["pred_ints"], C.c_int32), _p(f["pred_longs"], C.c_int64),
                                               len(f["pred_ints"]) // 4, _p(f["set_offsets"], C.c_int32), _p(f["set_words"], C.c_uint32), len(f["set_words"]),
                                               _p(f["aggregations"], C.c_int32), len(f["aggregations"]) // 2, _p(f["group_by"], C.c_int32), len(f["group_by"]),
                                               f["num_groups_limit"], f["flags"])
        if not self.handle:
            raise ValueError((self.lib.pgm_last_error() or b"").decode())
        self.c = self.lib.pgm_query_get(self.handle).contents

    def close(self):
        if self.handle:
            self.lib.pgm_query_free(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def unpack_result(res, is_group_by):
    """(header, group_ids, counts, sums, sums_i64, sum_exact, mins, maxs) of a pg_result, through pgm_result_header / pgm_result_fill."""
    lib = load()
    header = np.zeros(HEADER_LEN, dtype=np.int64)
    lib.pgm_result_header(C.byref(res), int(is_group_by), _p(header, C.c_int64))
    rows = int(lib.pgm_result_rows(C.byref(res), int(is_group_by)))
    cells = rows * int(header[H_NUM_AGGREGATIONS])
    group_ids = np.zeros(rows if is_group_by else 0, dtype=np.int32)
    counts, sums_i64 = np.zeros(cells, dtype=np.int64), np.zeros(cells, dtype=np.int64)
    sums, mins, maxs = np.zeros(cells, dtype=np.float64), np.zeros(cells, dtype=np.float64), np.zeros(cells, dtype=np.float64)
    sum_exact = np.zeros(cells, dtype=np.int32)
    wrote = lib.pgm_result_fill(C.byref(res), int(is_group_by), _p(group_ids, C.c_int32), _p(counts, C.c_int64), _p(sums, C.c_double), _p(sums_i64, C.c_int64),
                                _p(sum_exact, C.c_int32), _p(mins, C.c_double), _p(maxs, C.c_double))
    assert wrote == rows
    return header, group_ids, counts, sums, sums_i64, sum_exact, mins, maxs


def unpack_keys(res, num_group_by):
    """The dictId tuples of a group-by result ([rows, num_group_by]), through pgm_result_fill_keys."""
    lib = load()
    rows = int(res.num_groups)
    keys = np.zeros(rows * num_group_by, dtype=np.int32)
    wrote = lib.pgm_result_fill_keys(C.byref(res), int(num_group_by), _p(keys, C.c_int32))
    assert wrote == rows * num_group_by
    return keys.reshape(rows, num_group_by)
