"""Runs the JNI functions of jni/pinot_gpu_jni.c without a JVM: libpinot_gpu_jni_fake.so = pinot_gpu_jni.c + pg_marshal.c + fake_jvm.c
(an implementation of the JNIEnv functions of jni/stub/jni.h with the JNI specification's semantics) linked to the real libpinot_gpu.so.
Test infrastructure (tests/test_jni_harness_cpu.py, tests/test_gpu_jni_harness.py): what PinotGpuNative.java would call, called from
Python through ctypes -- array pinning, exception mapping, local-reference discipline and the batch call execute for real."""
import ctypes as C
import os

import numpy as np

from . import marshal as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "jni", "libpinot_gpu_jni_fake.so")
PREFIX = "Java_org_apache_pinot_gpu_PinotGpuNative_"
FJ_INT_ARRAY, FJ_LONG_ARRAY, FJ_DOUBLE_ARRAY, FJ_OBJECT_ARRAY, FJ_STRING = 1, 2, 3, 4, 5


def _header_constants():
    """The PGM_* enumerators of jni/pg_marshal.h (record sizes of the flat arrays): read from the header, not restated here."""
    import re
    text = open(os.path.join(ROOT, "jni", "pg_marshal.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return {name: int(value) for name, value in re.findall(r"\b(PGM_[A-Z0-9_]+)\s*=\s*(\d+)", text)}


_PGM = _header_constants()
COLUMN_INTS, COLUMN_BUFFERS, QUERY_ARRAYS = _PGM["PGM_COLUMN_INTS"], _PGM["PGM_COLUMN_BUFFERS"], _PGM["PGM_QUERY_ARRAYS"]


class JavaException(Exception):
    def __init__(self, cls, message):
        super().__init__("%s: %s" % (cls, message))
        self.cls = cls
        self.message = message


class FakeJvm:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("jni/libpinot_gpu_jni_fake.so is missing: run __graft_entry__.build() (make -C jni fake)")
        self.lib = lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        vp = C.c_void_p
        lib.fj_env.restype = vp
        for name, res, args in (("fj_int_array", vp, [vp, C.c_int32]), ("fj_long_array", vp, [vp, C.c_int32]), ("fj_object_array", vp, [C.c_int32]),
                                ("fj_string", vp, [C.c_char_p]), ("fj_set", None, [vp, C.c_int32, vp]), ("fj_get", vp, [vp, C.c_int32]),
                                ("fj_kind", C.c_int32, [vp]), ("fj_len", C.c_int32, [vp]), ("fj_data", vp, [vp]), ("fj_release", None, [vp]),
                                ("fj_exception_pending", C.c_int32, []), ("fj_exception_class", C.c_char_p, []), ("fj_exception_message", C.c_char_p, []),
                                ("fj_exception_clear", None, []), ("fj_live_refs", C.c_int64, []), ("fj_peak_refs", C.c_int64, []), ("fj_reset_peak", None, []),
                                ("fj_live_objects", C.c_int64, []), ("fj_pins", C.c_int64, []), ("fj_push_frame", None, []), ("fj_pop_frame", None, [vp])):
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        self.env = C.c_void_p(lib.fj_env())

    # ---- Java values ----
    def ints(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        return C.c_void_p(self.lib.fj_int_array(a.ctypes.data, a.shape[0]))

    def longs(self, a):
        a = np.ascontiguousarray(a, dtype=np.int64)
        return C.c_void_p(self.lib.fj_long_array(a.ctypes.data, a.shape[0]))

    def string(self, s):
        return C.c_void_p(self.lib.fj_string(s.encode()))

    def objects(self, items):
        """Object[] holding `items` (the array takes its own references; the items' local references are given back)."""
        arr = C.c_void_p(self.lib.fj_object_array(len(items)))
        for i, it in enumerate(items):
            self.lib.fj_set(arr, i, it)
            if it is not None and it.value:
                self.lib.fj_release(it)
        return arr

    def release(self, *objs):
        for o in objs:
            if o is not None and o.value:
                self.lib.fj_release(o)

    def to_python(self, obj):
        """A returned object as Python data (numpy copies); None for null."""
        if obj is None or not obj:
            return None
        obj = C.c_void_p(obj) if isinstance(obj, int) else obj
        kind, n, data = self.lib.fj_kind(obj), self.lib.fj_len(obj), self.lib.fj_data(obj)
        if kind == FJ_INT_ARRAY:
            return np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_int32)), (n,)).copy() if n else np.zeros(0, np.int32)
        if kind == FJ_LONG_ARRAY:
            return np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_int64)), (n,)).copy() if n else np.zeros(0, np.int64)
        if kind == FJ_DOUBLE_ARRAY:
            return np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_double)), (n,)).copy() if n else np.zeros(0, np.float64)
        if kind == FJ_STRING:
            return C.string_at(data).decode("utf-8", "replace")
        if kind == FJ_OBJECT_ARRAY:
            return [self.to_python(self.lib.fj_get(obj, i)) for i in range(n)]
        raise ValueError("object kind %d" % kind)

    # ---- native methods ----
    def call(self, name, restype, *args):
        """PinotGpuNative.<name>(args): raises JavaException when the native method left one pending."""
        fn = getattr(self.lib, PREFIX + name)
        fn.restype = restype
        fn.argtypes = None
        self.lib.fj_push_frame()
        out = fn(self.env, None, *args)
        self.lib.fj_pop_frame(out if restype is C.c_void_p else None)          # what the method left behind goes, as when a native method returns
        if self.lib.fj_exception_pending():
            cls, msg = self.lib.fj_exception_class().decode(), self.lib.fj_exception_message().decode("utf-8", "replace")
            self.lib.fj_exception_clear()
            raise JavaException(cls, msg)
        return out

    def query_arrays(self, spec):
        """The seven arrays GpuQueryLowering.Lowered holds (+ numGroupsLimit, flags), as Java objects."""
        f = M.flatten(spec)
        arrays = [self.ints(f["nodes"]), self.ints(f["pred_ints"]), self.longs(f["pred_longs"]), self.ints(f["set_offsets"]),
                  self.ints(f["set_words"].view(np.int32)), self.ints(f["aggregations"]), self.ints(f["group_by"])]
        return arrays, f["num_groups_limit"], f["flags"]

    def segment_open(self, seg, device=0):
        """PinotGpuNative.segmentOpen over a pinot_amd.segment.SegmentData (what GpuSegment.java does with the mapped index buffers)."""
        n = len(seg.columns)
        ints = np.zeros(COLUMN_INTS * n, dtype=np.int32)
        bufs = np.zeros(COLUMN_BUFFERS * n, dtype=np.int64)
        for i, c in enumerate(seg.columns):
            ints[COLUMN_INTS * i:COLUMN_INTS * i + COLUMN_INTS] = (c.stored_type, c.encoding, c.bits, c.cardinality, 0 if c.dictionary is None else 1, 0)
            b = [c.fwd.ctypes.data, c.fwd.nbytes, 0, 0, 0, 0, 0, 0]
            if c.dictionary is not None:
                b[2:4] = [c.dictionary.ctypes.data, c.dictionary.nbytes]
            if c.inverted is not None:
                b[4:6] = [c.inverted.ctypes.data, c.inverted.nbytes]
            if c.null_vector is not None:
                b[6:8] = [c.null_vector.ctypes.data, c.null_vector.nbytes]
            bufs[COLUMN_BUFFERS * i:COLUMN_BUFFERS * i + COLUMN_BUFFERS] = b
        name, names = self.string(seg.name), self.objects([self.string(c.name) for c in seg.columns])
        jints, jbufs = self.ints(ints), self.longs(bufs)
        try:
            return self.call("segmentOpen", C.c_int64, name, C.c_int64(0), C.c_int32(device), C.c_int32(seg.num_docs), names, jints, jbufs)
        finally:
            self.release(name, names, jints, jbufs)

    def execute(self, handle, spec):
        """PinotGpuNative.execute: the Object[PGM_RESULT_ARRAYS] as a list of numpy arrays."""
        arrays, limit, flags = self.query_arrays(spec)
        try:
            out = self.call("execute", C.c_void_p, C.c_int64(handle), *arrays, C.c_int32(limit), C.c_int32(flags))
            try:
                return self.to_python(out)
            finally:
                self.release(C.c_void_p(out))
        finally:
            self.release(*arrays)

    def query_check(self, handle, spec):
        arrays, limit, flags = self.query_arrays(spec)
        try:
            return self.call("queryCheck", C.c_int32, C.c_int64(handle), *arrays, C.c_int32(limit), C.c_int32(flags))
        finally:
            self.release(*arrays)

    def batch_queries(self, specs):
        items = []
        for spec in specs:
            arrays, limit, flags = self.query_arrays(spec)
            items.append(self.objects(arrays + [self.ints([limit, flags])]))
        return self.objects(items)

    def execute_batch(self, handles, specs):
        """PinotGpuNative.executeBatch: per item the list of result arrays, or the "<status>\\n<message>" string of a failed item."""
        jh, jq = self.longs(handles), self.batch_queries(specs)
        try:
            out = self.call("executeBatch", C.c_void_p, jh, jq)
            try:
                return self.to_python(out)
            finally:
                self.release(C.c_void_p(out))
        finally:
            self.release(jh, jq)
