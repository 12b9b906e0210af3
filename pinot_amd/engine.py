"""Thin Python handle over the C ABI (`libpinot_gpu.so`): what the JNI shim does, for tests and bench.py."""
import ctypes as C
import os

import numpy as np

from . import _abi
from .query import Result


class Engine:
    def __init__(self, device_id=0, time_kernels=False, blocks_per_cu=0, lib_path=None, profile_waves=False):
        self.lib = _abi.load_gpu_library(lib_path)
        flags = (_abi.PG_CFG_TIME_KERNELS if time_kernels else 0) | (_abi.PG_CFG_PROFILE_WAVES if profile_waves else 0)
        self._cfg = _abi.pg_config(_abi.PG_ABI_VERSION, device_id, blocks_per_cu, flags)
        _abi.check(self.lib, self.lib.pg_init(C.byref(self._cfg)))
        self.device_id = device_id

    def reinit(self, time_kernels=None, **env):
        """pg_init again with environment switches changed (value None = unset): the library re-reads them; open segments stay open.
        time_kernels: PG_CFG_TIME_KERNELS on / off from here on (None: as it was)."""
        if time_kernels is not None:
            flags = int(self._cfg.flags) & ~_abi.PG_CFG_TIME_KERNELS
            self._cfg.flags = flags | (_abi.PG_CFG_TIME_KERNELS if time_kernels else 0)
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        _abi.check(self.lib, self.lib.pg_init(C.byref(self._cfg)))

    def device_info(self):
        name = C.create_string_buffer(64)
        cus = C.c_int32()
        hbm = C.c_uint64()
        _abi.check(self.lib, self.lib.pg_device_info(self.device_id, name, 64, C.byref(cus), C.byref(hbm)))
        return name.value.decode(), int(cus.value), int(hbm.value)

    def device_count(self):
        """pg_device_count: (device ids accepted, HIP devices behind them) -- they differ under PINOT_GPU_ALIAS_DEVICES."""
        n, phys = C.c_int32(), C.c_int32()
        _abi.check(self.lib, self.lib.pg_device_count(C.byref(n), C.byref(phys)))
        return int(n.value), int(phys.value)

    def execute_batch(self, gsegs, specs):
        """pg_execute_batch: specs[i] over gsegs[i] (the same query lowered per segment).  Returns [(status, Result | None)]."""
        n = len(gsegs)
        handles = (C.c_void_p * max(n, 1))(*[g.handle for g in gsegs])
        queries = (C.POINTER(_abi.pg_query) * max(n, 1))(*[C.pointer(s.c) for s in specs])
        results = (_abi.pg_result * max(n, 1))()
        statuses = (C.c_int * max(n, 1))()
        _abi.check(self.lib, self.lib.pg_execute_batch(handles, queries, n, results, statuses))
        out = []
        for i in range(n):
            out.append((int(statuses[i]), Result(results[i], specs[i]) if statuses[i] == _abi.PG_OK else None))
            self.lib.pg_result_free(C.byref(results[i]))
        return out

    def execute_batch_raw(self, handles, queries, n, results, statuses):
        """Hot-loop variant for bench.py: ctypes arrays prepared by the caller, who frees the results."""
        return self.lib.pg_execute_batch(handles, queries, n, results, statuses)

    def open(self, segment_data):
        handle = C.c_void_p()
        _abi.check(self.lib, self.lib.pg_segment_open(C.byref(segment_data.desc), C.byref(handle)))
        return GpuSegment(self, handle, segment_data.num_docs)


class GpuSegment:
    def __init__(self, engine, handle, num_docs):
        self.engine = engine
        self.lib = engine.lib
        self.handle = handle
        self.num_docs = num_docs

    def close(self):
        if self.handle:
            _abi.check(self.lib, self.lib.pg_segment_close(self.handle))
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_bytes(self):
        out = C.c_uint64()
        _abi.check(self.lib, self.lib.pg_segment_device_bytes(self.handle, C.byref(out)))
        return int(out.value)

    def plane_bytes(self):
        out = C.c_uint64()
        _abi.check(self.lib, self.lib.pg_segment_plane_bytes(self.handle, C.byref(out)))
        return int(out.value)

    def execute(self, spec):
        res = _abi.pg_result()
        if os.environ.get("PINOT_GPU_ASSERT_QUERY_CHECK"):
            # the test suite's standing check of pg_query_check: a query it admits is never declined by pg_execute, and vice versa
            admitted = self.lib.pg_query_check(self.handle, C.byref(spec.c))
            status = self.lib.pg_execute(self.handle, C.byref(spec.c), C.byref(res))
            assert (admitted == _abi.PG_ERR_UNSUPPORTED) == (status == _abi.PG_ERR_UNSUPPORTED), "pg_query_check %d, pg_execute %d" % (admitted, status)
            _abi.check(self.lib, status)
        else:
            _abi.check(self.lib, self.lib.pg_execute(self.handle, C.byref(spec.c), C.byref(res)))
        try:
            return Result(res, spec)
        finally:
            self.lib.pg_result_free(C.byref(res))

    def group_key_info(self, column):
        """pg_group_key_info: (base, is_offset, null_entry) -- a raw INT / LONG group-by column's key value is base + its entry of
        Result.group_keys; null_entry is the entry that means NULL under null handling."""
        base, is_offset, null_entry = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        _abi.check(self.lib, self.lib.pg_group_key_info(self.handle, int(column), C.byref(base), C.byref(is_offset), C.byref(null_entry)))
        return int(base.value), int(is_offset.value), int(null_entry.value)

    def group_key_values(self, column, dtype=None):
        """pg_group_key_values: the distinct values of a raw group-by column keyed through a rank image (group_key_info's is_offset == 2),
        ascending -- np.int64 values, or np.float64 when `dtype` is a floating type (the bits are those of the double)."""
        n = C.c_int32(0)
        _abi.check(self.lib, self.lib.pg_group_key_values(self.handle, int(column), None, 0, C.byref(n)))
        out = np.zeros(max(int(n.value), 1), dtype=np.int64)
        _abi.check(self.lib, self.lib.pg_group_key_values(self.handle, int(column), out.ctypes.data_as(C.POINTER(C.c_int64)), int(out.shape[0]), C.byref(n)))
        out = out[: int(n.value)]
        return out.view(np.float64) if dtype is not None and np.issubdtype(np.dtype(dtype), np.floating) else out

    def check(self, spec):
        """pg_query_check: the status pg_execute would return for eligibility reasons (0 = PG_OK, 2 = PG_ERR_UNSUPPORTED).  Nothing is launched, except
        that the first check of a GROUP BY over a rank-keyed raw column builds its dictionary and rank image (a failed build: PG_ERR_UNSUPPORTED)."""
        return int(self.lib.pg_query_check(self.handle, C.byref(spec.c)))

    def execute_raw(self, spec, res):
        """Hot-loop variant for bench.py: no Python-side result conversion; caller frees `res`."""
        return self.lib.pg_execute(self.handle, C.byref(spec.c), C.byref(res))

    def filter_bitmap(self, spec):
        words = np.zeros((self.num_docs + 63) // 64, dtype=np.uint64)
        card = C.c_int64()
        _abi.check(self.lib, self.lib.pg_filter_bitmap(self.handle, C.byref(spec.c), words.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                       int(words.shape[0]), C.byref(card)))
        return words, int(card.value)

    def _read(self, fn, column, doc_ids, dtype, ctype):
        doc_ids = np.ascontiguousarray(doc_ids, dtype=np.int32)
        out = np.zeros(doc_ids.shape[0], dtype=dtype)
        _abi.check(self.lib, fn(self.handle, column, doc_ids.ctypes.data_as(C.POINTER(C.c_int32)), int(doc_ids.shape[0]),
                                out.ctypes.data_as(C.POINTER(ctype))))
        return out

    def read_dict_ids(self, column, doc_ids):
        return self._read(self.lib.pg_read_dict_ids, column, doc_ids, np.int32, C.c_int32)

    def read_int_values(self, column, doc_ids):
        return self._read(self.lib.pg_read_int_values, column, doc_ids, np.int32, C.c_int32)

    def read_long_values(self, column, doc_ids):
        return self._read(self.lib.pg_read_long_values, column, doc_ids, np.int64, C.c_int64)

    def read_double_values(self, column, doc_ids):
        return self._read(self.lib.pg_read_double_values, column, doc_ids, np.float64, C.c_double)
