"""Python handle over the C++ host mirror (libpinot_host.so): SQL in, results blocks out.

Plays the role of the reference's test harness `BaseQueriesTest.getOperator(sql)` / `getBrokerResponse(sql, planMaker)`
(pinot-core/src/test/java/org/apache/pinot/queries/BaseQueriesTest.java:100-105,154-156): the query is parsed, lowered
with dictionary binary searches and planned by the C++ `GpuPlanMaker`, executed through the C ABI on the device, and the
per-segment blocks are merged by the C++ combine step.
"""
import ctypes as C
import json

from . import _abi
from .segment import load_host_library


class HostError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status   # 1 bad query, 2 not offloadable (keep the CPU plan), 3 runtime


def _lib():
    lib = load_host_library()
    if getattr(lib, "_host_bound", False):
        return lib
    vp = C.c_void_p
    lib.ph_last_error.restype = C.c_char_p
    lib.ph_free.argtypes = [vp]
    lib.ph_segment_create.restype = vp
    lib.ph_segment_create.argtypes = [C.c_char_p, C.c_int32]
    lib.ph_segment_add_int_column.restype = C.c_int32
    lib.ph_segment_add_int_column.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64]
    lib.ph_segment_add_numeric_column.restype = C.c_int32
    lib.ph_segment_add_numeric_column.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64]
    lib.ph_segment_add_string_column.restype = C.c_int32
    lib.ph_segment_add_string_column.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, vp, C.c_uint64, C.c_char_p, vp, C.c_uint64]
    lib.ph_segment_set_null_vector.restype = C.c_int32
    lib.ph_segment_set_null_vector.argtypes = [vp, C.c_char_p, vp, C.c_uint64]
    lib.ph_segment_load.restype = C.c_int32
    lib.ph_segment_load.argtypes = [vp, C.c_int32]
    lib.ph_segment_destroy.argtypes = [vp]
    lib.ph_plan_maker_init.restype = C.c_int32
    lib.ph_plan_maker_init.argtypes = [C.c_int32, C.c_int32]
    lib.ph_segment_load_directory.restype = vp
    lib.ph_segment_load_directory.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]
    lib.ph_segment_describe.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.ph_plan_maker_placement.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32)]
    for name in ("ph_parse_sql", "ph_lower_predicate", "ph_execute_sql", "ph_segment_describe", "ph_plan_maker_placement"):
        getattr(lib, name).restype = vp
    lib.ph_parse_sql.argtypes = [C.c_char_p, C.POINTER(C.c_int32)]
    lib.ph_lower_predicate.argtypes = [C.c_char_p, vp, C.c_int32, C.POINTER(C.c_int32)]
    lib.ph_execute_sql.argtypes = [C.POINTER(vp), C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]
    lib._host_bound = True
    return lib


def _take_json(lib, ptr, status):
    if status.value != 0 or not ptr:
        raise HostError(status.value, (lib.ph_last_error() or b"").decode("utf-8", "replace"))
    try:
        return json.loads(C.string_at(ptr).decode("utf-8"))
    finally:
        lib.ph_free(ptr)


def parse_sql(sql):
    lib = _lib()
    st = C.c_int32()
    return _take_json(lib, lib.ph_parse_sql(sql.encode(), C.byref(st)), st)


def lower_predicate(predicate_sql, dictionary_bytes, cardinality):
    lib = _lib()
    st = C.c_int32()
    return _take_json(lib, lib.ph_lower_predicate(predicate_sql.encode(), dictionary_bytes.ctypes.data, cardinality, C.byref(st)), st)


def placement(devices_text, segment_bytes):
    """GpuPlanMaker's gpu.devices parsing + least-loaded placement (no device touched): {"devices": [...], "placement": [...]}.
    A negative entry -(device << 48 | bytes) gives bytes back (placement -1)."""
    lib = _lib()
    st = C.c_int32()
    arr = (C.c_int64 * max(len(segment_bytes), 1))(*segment_bytes)
    return _take_json(lib, lib.ph_plan_maker_placement(devices_text.encode(), arr, len(segment_bytes), C.byref(st)), st)


def init_plan_maker(device=0, time_kernels=True):
    lib = _lib()
    st = lib.ph_plan_maker_init(device, int(time_kernels))
    if st != 0:
        raise HostError(st, (lib.ph_last_error() or b"").decode())


def _attach_null_vectors(lib, handle, columns):
    for c in columns:
        if getattr(c, "null_vector", None) is not None:
            st = lib.ph_segment_set_null_vector(handle, c.name.encode(), c.null_vector.ctypes.data, c.null_vector.nbytes)
            if st != 0:
                raise HostError(st, lib.ph_last_error().decode())


class HostSegment:
    """ImmutableSegment of the C++ host mirror built from a `SegmentData` (buffers stay owned by the SegmentData)."""

    def __init__(self, segment_data, string_dicts=None, device=0, load=True):
        """load=False: the host-side segment only (data sources, dictionaries, sorted / inverted index metadata), nothing on a device --
        enough for parse / plan inspection (`explain_filter`)."""
        lib = _lib()
        self.lib = lib
        self.data = segment_data
        self.handle = C.c_void_p(lib.ph_segment_create(segment_data.name.encode(), segment_data.num_docs))
        string_dicts = string_dicts or {}
        for c in segment_data.columns:
            inv_ptr = c.inverted.ctypes.data if c.inverted is not None else None
            inv_size = c.inverted.nbytes if c.inverted is not None else 0
            if c.name in string_dicts:
                values = b"".join(s.encode("utf-8") + b"\0" for s in string_dicts[c.name])
                st = lib.ph_segment_add_string_column(self.handle, c.name.encode(), c.bits, c.cardinality, c.fwd.ctypes.data, c.fwd.nbytes,
                                                      values, inv_ptr, inv_size)
            else:
                has_dict = c.encoding == _abi.PG_FWD_FIXED_BIT_DICT
                st = lib.ph_segment_add_numeric_column(self.handle, c.name.encode(), c.stored_type, int(has_dict), c.bits, c.cardinality,
                                                       c.fwd.ctypes.data, c.fwd.nbytes, c.dictionary.ctypes.data if has_dict else None,
                                                       c.dictionary.nbytes if has_dict else 0, inv_ptr, inv_size)
            if st != 0:
                raise HostError(st, (lib.ph_last_error() or b"").decode())
        _attach_null_vectors(lib, self.handle, segment_data.columns)
        if load:
            st = lib.ph_segment_load(self.handle, device)
            if st != 0:
                raise HostError(st, (lib.ph_last_error() or b"").decode())

    def destroy(self):
        if self.handle:
            self.lib.ph_segment_destroy(self.handle)
            self.handle = None


class DirectorySegment:
    """A v1 / v3 Pinot segment directory opened by the native loader (ImmutableSegmentLoader.load) and made HBM resident."""

    def __init__(self, index_dir, device=0):
        lib = _lib()
        self.lib = lib
        st = C.c_int32()
        self.handle = C.c_void_p(lib.ph_segment_load_directory(str(index_dir).encode(), device, C.byref(st)))
        if st.value != 0 or not self.handle:
            self.handle = None
            raise HostError(st.value, (lib.ph_last_error() or b"").decode("utf-8", "replace"))

    def describe(self):
        st = C.c_int32()
        return _take_json(self.lib, self.lib.ph_segment_describe(self.handle, C.byref(st)), st)

    def destroy(self):
        if self.handle:
            self.lib.ph_segment_destroy(self.handle)
            self.handle = None


def explain_filter(segment, sql):
    """The physical filter operator tree (FilterPlanNode + FilterOperatorUtils) of `sql`'s WHERE clause over `segment`, as text."""
    lib = _lib()
    lib.ph_explain_filter.restype = C.c_void_p
    lib.ph_explain_filter.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32)]
    st = C.c_int32()
    ptr = lib.ph_explain_filter(segment.handle, sql.encode(), C.byref(st))
    if st.value != 0 or not ptr:
        raise HostError(st.value, (lib.ph_last_error() or b"").decode("utf-8", "replace"))
    try:
        return C.string_at(ptr).decode("utf-8")
    finally:
        lib.ph_free(ptr)


def execute_sql(segments, sql, max_execution_threads=0):
    """Returns {"segments": [per-segment block...], "combined": block}."""
    lib = _lib()
    arr = (C.c_void_p * len(segments))(*[s.handle for s in segments])
    st = C.c_int32()
    return _take_json(lib, lib.ph_execute_sql(arr, len(segments), sql.encode(), max_execution_threads, C.byref(st)), st)


KEY_INT, KEY_LONG, KEY_FLOAT, KEY_DOUBLE, KEY_STRING = range(5)      # DataType ordinals of the key arrays (host/c_api.cpp)


def group_by_combine(sql, blocks, key_types):
    """GroupByCombineOperator + GroupByDataTableReducer over group-by blocks built on the host (no device): the IndexedTable / TableResizer
    mirror of pinot_amd/csrc/host/indexed_table.cpp.  `blocks`: one list of rows per segment, a row = (key values, cells) with one cell
    (count, sum, min, max, is_null) per aggregation of `sql`, None as a key value = NULL.  Returns {"combined", "reduced", "table"}."""
    lib = _lib()
    P = C.POINTER
    lib.ph_group_by_combine.restype = C.c_void_p
    lib.ph_group_by_combine.argtypes = [C.c_char_p, C.c_int32, P(C.c_int64), P(C.c_int32), P(C.c_int64), P(C.c_double), P(C.c_char_p), P(C.c_uint8), P(C.c_int64),
                                        P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_uint8), P(C.c_int32)]
    rows = [r for b in blocks for r in b]
    nk = len(key_types)
    nf = len(rows[0][1]) if rows else 1
    nr = len(rows)
    br = (C.c_int64 * max(len(blocks), 1))(*[len(b) for b in blocks])
    kt = (C.c_int32 * max(nk, 1))(*key_types)
    kl, kd, ks, kn = (C.c_int64 * max(nr * nk, 1))(), (C.c_double * max(nr * nk, 1))(), (C.c_char_p * max(nr * nk, 1))(), (C.c_uint8 * max(nr * nk, 1))()
    counts, sums = (C.c_int64 * max(nr * nf, 1))(), (C.c_double * max(nr * nf, 1))()
    mins, maxs, nulls = (C.c_double * max(nr * nf, 1))(), (C.c_double * max(nr * nf, 1))(), (C.c_uint8 * max(nr * nf, 1))()
    for r, (key_values, cells) in enumerate(rows):
        for k, v in enumerate(key_values):
            at = r * nk + k
            if v is None:
                kn[at] = 1
                ks[at] = b""
            elif key_types[k] in (KEY_INT, KEY_LONG):
                kl[at] = int(v)
            elif key_types[k] == KEY_STRING:
                ks[at] = str(v).encode()
            else:
                kd[at] = float(v)
        for f, (c, s, mn, mx, is_null) in enumerate(cells):
            at = r * nf + f
            counts[at], sums[at], mins[at], maxs[at], nulls[at] = int(c), float(s), float(mn), float(mx), int(bool(is_null))
    st = C.c_int32()
    return _take_json(lib, lib.ph_group_by_combine(sql.encode(), len(blocks), br, kt, kl, kd, ks, kn, counts, sums, mins, maxs, nulls, C.byref(st)), st)


def execute_sql_datatable(segments, sql, max_execution_threads=0):
    """The DataTable V4 bytes a server would send the broker for `sql` over these segments (combine, then
    InstanceResponseBlock.toDataTable().toBytes(); pinot_amd/csrc/host/datatable_v4.cpp)."""
    lib = _lib()
    lib.ph_execute_sql_datatable.restype = C.POINTER(C.c_uint8)
    lib.ph_execute_sql_datatable.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    handles = (C.c_void_p * len(segments))(*[s.handle for s in segments])
    size, status = C.c_int64(), C.c_int32()
    ptr = lib.ph_execute_sql_datatable(handles, len(segments), sql.encode(), max_execution_threads, C.byref(size), C.byref(status))
    if status.value != 0 or not ptr:
        raise HostError(status.value, (lib.ph_last_error() or b"").decode("utf-8", "replace"))
    data = bytes(bytearray(ptr[:size.value]))
    lib.ph_free.argtypes = [C.c_void_p]
    lib.ph_free(C.cast(ptr, C.c_void_p))
    return data
