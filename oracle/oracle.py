"""ctypes binding of oracle/_build/libpinot_oracle.so (the CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
It consumes the same `pg_segment_desc` / `pg_query` PODs as the engine so one query description drives both.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from pinot_amd import _abi
from pinot_amd.query import Result

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "_build", "libpinot_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_DIR, "pinot_oracle.c")
    hdr = os.path.join(_DIR, "..", "include", "pinot_gpu.h")
    stale = (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return _LIB


def load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    lib = C.CDLL(_LIB)
    P = C.POINTER
    u8p, i32p = P(C.c_uint8), P(C.c_int32)
    sigs = {
        "po_last_error": (C.c_char_p, []),
        "po_num_bits_per_value": (C.c_int, [C.c_int32]),
        "po_bitset_read_int": (C.c_int32, [u8p, C.c_int64, C.c_int]),
        "po_bitset_write_int": (None, [u8p, C.c_int64, C.c_int, C.c_int32]),
        "po_fixedbit_file_size": (C.c_int64, [C.c_int64, C.c_int]),
        "po_fixedbit_write": (None, [u8p, i32p, C.c_int64, C.c_int]),
        "po_fixedbit_read_dict_ids": (None, [u8p, C.c_int, C.c_int32, i32p, C.c_int32, i32p]),
        "po_raw_file_size_v2": (C.c_int64, [C.c_int32, C.c_int32]),
        "po_raw_write_int_v2": (None, [u8p, i32p, C.c_int32, C.c_int32]),
        "po_dict_get_int": (C.c_int32, [u8p, C.c_int32]),
        "po_dict_insertion_index_of_int": (C.c_int32, [u8p, C.c_int32, C.c_int32]),
        "po_dict_index_of_int": (C.c_int32, [u8p, C.c_int32, C.c_int32]),
        "po_dict_write_int": (None, [u8p, i32p, C.c_int32]),
        "po_lower_range_int": (None, [u8p, C.c_int32, C.c_int, C.c_int32, C.c_int, C.c_int, C.c_int32, C.c_int, i32p, i32p]),
        "po_dict_write_long": (None, [u8p, P(C.c_int64), C.c_int32]),
        "po_dict_write_float": (None, [u8p, P(C.c_float), C.c_int32]),
        "po_dict_write_double": (None, [u8p, P(C.c_double), C.c_int32]),
        "po_dict_insertion_index_of": (C.c_int32, [u8p, C.c_int32, C.c_int, C.c_int64, C.c_double]),
        "po_lower_range_typed": (None, [u8p, C.c_int32, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_int, i32p, i32p]),
        "po_raw_file_size_typed_v2": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
        "po_raw_write_typed_v2": (None, [u8p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
        "po_raw_header": (C.c_int, [u8p, C.c_uint64, i32p]),
        "po_read_double_values": (C.c_int, [P(_abi.pg_segment_desc), C.c_int32, i32p, C.c_int32, P(C.c_double), P(C.c_int64)]),
        "po_roaring_or_into": (C.c_int64, [u8p, C.c_uint64, P(C.c_uint64), C.c_int64]),
        "po_roaring_serialize": (C.c_int64, [i32p, C.c_int64, C.c_int, u8p]),
        "po_inverted_build": (C.c_int64, [i32p, C.c_int32, C.c_int32, C.c_int, u8p]),
        "po_execute": (C.c_int, [P(_abi.pg_segment_desc), P(_abi.pg_query), P(_abi.pg_result)]),
        "po_result_free": (None, [P(_abi.pg_result)]),
        "po_filter_bitmap": (C.c_int, [P(_abi.pg_segment_desc), P(_abi.pg_query), P(C.c_uint64), C.c_int64, P(C.c_int64)]),
        "po_read_int_values": (C.c_int, [P(_abi.pg_segment_desc), C.c_int32, i32p, C.c_int32, i32p]),
        "po_not_iterator_script": (C.c_int, [C.c_int, P(P(C.c_uint64)), C.c_int, C.c_int32, i32p, C.c_int, i32p, P(C.c_int64)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise OracleError("oracle rc=%d: %s" % (rc, (load().po_last_error() or b"").decode()))


def execute(segment_data, spec):
    lib = load()
    res = _abi.pg_result()
    _check(lib.po_execute(C.byref(segment_data.desc), C.byref(spec.c), C.byref(res)))
    try:
        return Result(res, spec)
    finally:
        lib.po_result_free(C.byref(res))


def execute_raw(segment_data, spec, res):
    return load().po_execute(C.byref(segment_data.desc), C.byref(spec.c), C.byref(res))


def filter_bitmap(segment_data, spec):
    lib = load()
    words = np.zeros((segment_data.num_docs + 63) // 64, dtype=np.uint64)
    card = C.c_int64()
    _check(lib.po_filter_bitmap(C.byref(segment_data.desc), C.byref(spec.c), words.ctypes.data_as(C.POINTER(C.c_uint64)),
                                int(words.shape[0]), C.byref(card)))
    return words, int(card.value)


def not_iterator_script(kind, members, num_docs, script):
    """Test hook (pinot_oracle.c po_not_iterator_script): a NotDocIdIterator over a bitmap (kind 0), an OrDocIdIterator of bitmaps (1) or
    a scan leaf (2) whose docId sets are `members` (lists of docIds), driven by `script` (-1: next(), t >= 0: advance(t)).
    Returns (docIds returned, entries the scan leaf counted)."""
    lib = load()
    nw = max(1, (num_docs + 63) // 64)
    keep = []
    ptrs = (C.POINTER(C.c_uint64) * len(members))()
    for i, docs in enumerate(members):
        w = np.zeros(nw, dtype=np.uint64)
        for d in docs:
            w[d >> 6] |= np.uint64(1) << np.uint64(d & 63)
        keep.append(w)
        ptrs[i] = w.ctypes.data_as(C.POINTER(C.c_uint64))
    script = np.ascontiguousarray(script, dtype=np.int32)
    out = np.zeros(script.shape[0], dtype=np.int32)
    entries = C.c_int64()
    _check(lib.po_not_iterator_script(kind, ptrs, len(members), num_docs, _i32p(script), int(script.shape[0]), _i32p(out), C.byref(entries)))
    return [int(x) for x in out], int(entries.value)


def read_int_values(segment_data, column, doc_ids):
    doc_ids = np.ascontiguousarray(doc_ids, dtype=np.int32)
    out = np.zeros(doc_ids.shape[0], dtype=np.int32)
    _check(load().po_read_int_values(C.byref(segment_data.desc), column, _i32p(doc_ids), int(doc_ids.shape[0]), _i32p(out)))
    return out


def read_double_values(segment_data, column, doc_ids):
    """(getDoubleValuesSV, getLongValuesSV) of any numeric column."""
    doc_ids = np.ascontiguousarray(doc_ids, dtype=np.int32)
    out = np.zeros(doc_ids.shape[0], dtype=np.float64)
    out_long = np.zeros(doc_ids.shape[0], dtype=np.int64)
    _check(load().po_read_double_values(C.byref(segment_data.desc), column, _i32p(doc_ids), int(doc_ids.shape[0]),
                                        out.ctypes.data_as(C.POINTER(C.c_double)), out_long.ctypes.data_as(C.POINTER(C.c_int64))))
    return out, out_long


def raw_header(buf):
    """(version, numChunks, numDocsPerChunk, sizeOfEntry, totalDocs, compressionType, dataHeaderStart, rawDataStart)"""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    out = np.zeros(8, dtype=np.int32)
    _check(load().po_raw_header(_u8p(buf), buf.nbytes, _i32p(out)))
    return tuple(int(x) for x in out)


def dict_write_typed(sorted_values):
    lib = load()
    v = np.ascontiguousarray(sorted_values)
    out = np.zeros(v.nbytes, dtype=np.uint8)
    if v.dtype == np.int32:
        lib.po_dict_write_int(_u8p(out), _i32p(v), int(v.shape[0]))
    elif v.dtype == np.int64:
        lib.po_dict_write_long(_u8p(out), v.ctypes.data_as(C.POINTER(C.c_int64)), int(v.shape[0]))
    elif v.dtype == np.float32:
        lib.po_dict_write_float(_u8p(out), v.ctypes.data_as(C.POINTER(C.c_float)), int(v.shape[0]))
    else:
        lib.po_dict_write_double(_u8p(out), v.ctypes.data_as(C.POINTER(C.c_double)), int(v.shape[0]))
    return out


def raw_write_typed(values, docs_per_chunk=1000):
    lib = load()
    v = np.ascontiguousarray(values)
    out = np.zeros(int(lib.po_raw_file_size_typed_v2(int(v.shape[0]), docs_per_chunk, v.dtype.itemsize)), dtype=np.uint8)
    lib.po_raw_write_typed_v2(_u8p(out), v.ctypes.data, int(v.shape[0]), docs_per_chunk, v.dtype.itemsize)
    return out


def lower_range_typed(column, lower=None, lower_inclusive=True, upper=None, upper_inclusive=True):
    """[startDictId, endDictId) of a range predicate on a dictionary column of any numeric stored type (bounds are numbers
    as Long.parseLong / Float.parseFloat / Double.parseDouble would return them)."""
    s, e = C.c_int32(), C.c_int32()
    integral = column.stored_type in (_abi.PG_TYPE_INT, _abi.PG_TYPE_LONG)
    lo_i, lo_d = (int(lower), 0.0) if (lower is not None and integral) else (0, float(lower or 0.0))
    hi_i, hi_d = (int(upper), 0.0) if (upper is not None and integral) else (0, float(upper or 0.0))
    load().po_lower_range_typed(_u8p(column.dictionary), column.cardinality, column.stored_type, int(lower is not None), lo_i, lo_d,
                                int(lower_inclusive), int(upper is not None), hi_i, hi_d, int(upper_inclusive), C.byref(s), C.byref(e))
    return int(s.value), int(e.value)


def read_dict_ids(fwd, bits, num_docs, doc_ids):
    doc_ids = np.ascontiguousarray(doc_ids, dtype=np.int32)
    out = np.zeros(doc_ids.shape[0], dtype=np.int32)
    # the reference's readUnchecked may read up to 7 bytes past a value that is not one of the last two: pad
    padded = np.concatenate([fwd, np.zeros(8, dtype=np.uint8)])
    load().po_fixedbit_read_dict_ids(_u8p(padded), bits, num_docs, _i32p(doc_ids), int(doc_ids.shape[0]), _i32p(out))
    return out


def fixedbit_write(dict_ids, bits):
    lib = load()
    dict_ids = np.ascontiguousarray(dict_ids, dtype=np.int32)
    out = np.zeros(int(lib.po_fixedbit_file_size(int(dict_ids.shape[0]), bits)), dtype=np.uint8)
    lib.po_fixedbit_write(_u8p(out), _i32p(dict_ids), int(dict_ids.shape[0]), bits)
    return out


def dict_write(sorted_values):
    v = np.ascontiguousarray(sorted_values, dtype=np.int32)
    out = np.zeros(v.shape[0] * 4, dtype=np.uint8)
    load().po_dict_write_int(_u8p(out), _i32p(v), int(v.shape[0]))
    return out


def raw_write(values, docs_per_chunk=1000):
    lib = load()
    v = np.ascontiguousarray(values, dtype=np.int32)
    out = np.zeros(int(lib.po_raw_file_size_v2(int(v.shape[0]), docs_per_chunk)), dtype=np.uint8)
    lib.po_raw_write_int_v2(_u8p(out), _i32p(v), int(v.shape[0]), docs_per_chunk)
    return out


def inverted_build(dict_ids, cardinality, run_optimize=True):
    lib = load()
    d = np.ascontiguousarray(dict_ids, dtype=np.int32)
    size = int(lib.po_inverted_build(_i32p(d), int(d.shape[0]), cardinality, int(run_optimize), None))
    out = np.zeros(size, dtype=np.uint8)
    lib.po_inverted_build(_i32p(d), int(d.shape[0]), cardinality, int(run_optimize), _u8p(out))
    return out


def lower_range(dictionary_bytes, cardinality, lower=None, upper=None, lower_inclusive=True, upper_inclusive=True):
    """SortedDictionaryBasedRangePredicateEvaluator bounds -> (startDictId, endDictId)."""
    s, e = C.c_int32(), C.c_int32()
    load().po_lower_range_int(_u8p(dictionary_bytes), cardinality, int(lower is not None), int(lower or 0), int(lower_inclusive),
                              int(upper is not None), int(upper or 0), int(upper_inclusive), C.byref(s), C.byref(e))
    return int(s.value), int(e.value)


def index_of(dictionary_bytes, cardinality, value):
    return int(load().po_dict_index_of_int(_u8p(dictionary_bytes), cardinality, int(value)))


def roaring_to_words(data, num_words):
    words = np.zeros(num_words, dtype=np.uint64)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    card = load().po_roaring_or_into(_u8p(data), int(data.shape[0]), words.ctypes.data_as(C.POINTER(C.c_uint64)), num_words)
    if card < 0:
        raise OracleError((load().po_last_error() or b"").decode())
    return words, int(card)


def execute_sliced(segment_data, spec, threads=None):
    """The oracle on every host core: the reference runs one segment per thread (BaseCombineOperator), so a big segment is cut into
    equal row ranges that start on multiples of 8 docs (8 docs of a b-bit column are b whole bytes: the slices are views of the same
    packed buffers and share the dictionaries, so raw group ids coincide), one oracle thread per slice, and the partial results are
    merged like AggregationFunction.merge does (SUM / COUNT '+', MIN / MAX min / max, AVG pairwise).  Checker plumbing only: used by
    bench.py to verify 1 B-row results in well under a second.  Dictionary-encoded columns and scan leaves only."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    from pinot_amd import segment as S
    n = segment_data.num_docs
    threads = threads or (os.cpu_count() or 1)
    parts = max(1, min(threads, n // 65536))
    for c in segment_data.columns:
        if c.encoding != _abi.PG_FWD_FIXED_BIT_DICT:
            parts = 1
    bounds = [min(n, ((n * i // parts) + 7) // 8 * 8) for i in range(parts)] + [n]
    slices = []
    for i in range(parts):
        lo, hi = bounds[i], bounds[i + 1]
        if hi <= lo:
            continue
        if parts == 1:
            slices.append(segment_data)
            continue
        cols = []
        for c in segment_data.columns:
            first = lo * c.bits // 8
            cols.append(S.Column(c.name, c.encoding, c.bits, c.cardinality, c.fwd[first:first + ((hi - lo) * c.bits + 7) // 8], c.dictionary, None, c.dict_values,
                                 stored_type=c.stored_type))
        slices.append(S.SegmentData("slice%d" % i, hi - lo, cols))

    def run(part):
        return execute(part, spec)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=max(1, len(slices))) as pool:
        results = list(pool.map(run, slices))
    seconds = time.perf_counter() - t0

    def fold(values):
        out = {"count": 0, "sum_i64": 0, "sum": 0.0, "min": float("inf"), "max": float("-inf")}
        for v in values:
            out["count"] += v.count
            out["sum_i64"] += v.sum_i64
            out["sum"] += v.sum
            out["min"] = min(out["min"], v.min)
            out["max"] = max(out["max"], v.max)
        return out
    na = len(spec.aggregations)
    merged = {"seconds": seconds, "threads": threads, "slices": len(slices), "docs_scanned": sum(r.stats[0] for r in results),
              "aggregations": [fold([r.aggregations[a] for r in results]) for a in range(na)] if not spec.group_by else [], "groups": {}}
    if spec.group_by:
        keys = set()
        for r in results:
            keys.update(r.groups)
        for g in keys:
            merged["groups"][g] = [fold([r.groups[g][a] for r in results if g in r.groups]) for a in range(na)]
    return merged


def execute_prefix(segment_data, spec, rows):
    """The oracle on ONE host core over the first `rows` docs of a segment (rounded down to a multiple of 8 docs: whole bytes of every
    bit width; the slice is a view of the same packed buffers) -- a bounded cpu_baseline sample of a 1 B-row workload.  Segments with
    raw columns or inverted indexes are not cut: the whole segment runs.  Returns (Result, seconds, rows that ran)."""
    import time
    from pinot_amd import segment as S
    n = segment_data.num_docs
    rows = min(n, max(8, rows // 8 * 8))
    cut = rows < n and all(c.encoding == _abi.PG_FWD_FIXED_BIT_DICT and c.inverted is None for c in segment_data.columns)
    part = segment_data
    if cut:
        cols = [S.Column(c.name, c.encoding, c.bits, c.cardinality, c.fwd[:(rows * c.bits + 7) // 8], c.dictionary, None, c.dict_values, stored_type=c.stored_type)
                for c in segment_data.columns]
        part = S.SegmentData("prefix", rows, cols)
    t0 = time.perf_counter()
    res = execute(part, spec)
    return res, time.perf_counter() - t0, part.num_docs


def matches_sliced(got, want, functions):
    """A pinot_amd.query.Result against execute_sliced's merge: bit exact on what each function defines."""
    def same(a, w, f):
        from pinot_amd import query as Q
        if a.count != w["count"]:
            return False
        if f in (Q.SUM, Q.AVG) and a.sum_i64 != w["sum_i64"]:
            return False
        if f == Q.MIN and a.min != w["min"]:
            return False
        if f == Q.MAX and a.max != w["max"]:
            return False
        return True
    if want["groups"] or got.groups:
        if sorted(got.groups) != sorted(want["groups"]):
            return False
        return all(same(got.groups[g][a], want["groups"][g][a], f) for g in want["groups"] for a, f in enumerate(functions))
    return all(same(got.aggregations[a], want["aggregations"][a], f) for a, f in enumerate(functions))
