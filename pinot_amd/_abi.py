"""ctypes mirror of include/pinot_gpu.h and the loader of the in-tree HIP library.

This module is plumbing for tests and bench.py (there is no JVM in this environment, so Python plays the
role of the JNI shim described in INTEGRATION.md).  It never computes anything: every call goes through the
C ABI into the HIP kernels, and loading fails loudly when the library has not been built.
"""
import ctypes as C
import os

PG_ABI_VERSION = 3

# pg_status
PG_OK, PG_ERR_INVALID_ARGUMENT, PG_ERR_UNSUPPORTED, PG_ERR_DEVICE, PG_ERR_OUT_OF_MEMORY, PG_ERR_NOT_INITIALIZED, PG_ERR_INTERNAL = range(7)
# pg_data_type / pg_fwd_encoding
KERNEL_NAMES = {0: "scan_agg_kernel", 1: "scan_private_kernel", 2: "scan_group_kernel", 3: "group_private_kernel",
                4: "group_partition_scatter_kernel", 5: "scan_private_typed_kernel", 6: "scan_hist_kernel", 7: "index_and_kernel", 8: "scan_narrow_kernel", 9: "scan_sparse_kernel", 10: "scan_simple_kernel", 11: "scan_raw_kernel"}
PG_TYPE_INT, PG_TYPE_LONG, PG_TYPE_FLOAT, PG_TYPE_DOUBLE = range(4)
PG_FWD_FIXED_BIT_DICT, PG_FWD_RAW_FIXED_BYTE = 0, 1
# pg_predicate_kind / pg_leaf_eval
PG_PRED_MATCH_ALL, PG_PRED_MATCH_NONE, PG_PRED_DICT_RANGE, PG_PRED_DICT_SET, PG_PRED_RAW_RANGE, PG_PRED_DOC_RANGE, PG_PRED_IS_NULL = range(7)
PG_QUERY_DEFAULT, PG_QUERY_NULL_HANDLING, PG_QUERY_STATS_UPPER_BOUND_OK = 0, 1, 2
PG_EVAL_SCAN, PG_EVAL_INVERTED = 0, 1
# pg_filter_op
PG_FILTER_LEAF, PG_FILTER_AND, PG_FILTER_OR, PG_FILTER_NOT = range(4)
# pg_agg_function
PG_AGG_COUNT, PG_AGG_SUM, PG_AGG_MIN, PG_AGG_MAX, PG_AGG_AVG = range(5)
PG_CFG_TIME_KERNELS = 1
PG_CFG_PROFILE_WAVES = 2


class pg_config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device_id", C.c_int32), ("blocks_per_cu", C.c_int32), ("flags", C.c_int32),
                ("plane_budget_bytes", C.c_uint64)]


class pg_column_desc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("stored_type", C.c_int32), ("fwd_encoding", C.c_int32),
                ("bits_per_value", C.c_int32), ("cardinality", C.c_int32),
                ("fwd_data", C.c_void_p), ("fwd_size", C.c_uint64),
                ("dict_data", C.c_void_p), ("dict_size", C.c_uint64),
                ("inv_data", C.c_void_p), ("inv_size", C.c_uint64),
                ("null_data", C.c_void_p), ("null_size", C.c_uint64)]


class pg_segment_desc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("crc", C.c_uint64), ("num_docs", C.c_int32), ("num_columns", C.c_int32),
                ("columns", C.POINTER(pg_column_desc)), ("device_id", C.c_int32), ("reserved", C.c_int32)]


class pg_predicate(C.Structure):
    _fields_ = [("kind", C.c_int32), ("column", C.c_int32), ("eval", C.c_int32), ("exclusive", C.c_int32),
                ("lo", C.c_int64), ("hi", C.c_int64), ("set_words", C.POINTER(C.c_uint32)),
                ("num_set_words", C.c_int32), ("reserved", C.c_int32)]


class pg_filter_node(C.Structure):
    _fields_ = [("op", C.c_int32), ("predicate", C.c_int32), ("num_children", C.c_int32), ("reserved", C.c_int32)]


class pg_aggregation(C.Structure):
    _fields_ = [("function", C.c_int32), ("column", C.c_int32)]


class pg_query(C.Structure):
    _fields_ = [("filter", C.POINTER(pg_filter_node)), ("num_filter_nodes", C.c_int32), ("num_predicates", C.c_int32),
                ("predicates", C.POINTER(pg_predicate)), ("aggregations", C.POINTER(pg_aggregation)),
                ("num_aggregations", C.c_int32), ("num_group_by", C.c_int32),
                ("group_by_columns", C.POINTER(C.c_int32)), ("num_groups_limit", C.c_int32), ("flags", C.c_int32)]


class pg_agg_value(C.Structure):
    _fields_ = [("count", C.c_int64), ("sum", C.c_double), ("sum_i64", C.c_int64), ("sum_exact", C.c_int32),
                ("reserved", C.c_int32), ("min", C.c_double), ("max", C.c_double)]


class pg_stats(C.Structure):
    _fields_ = [("num_docs_scanned", C.c_int64), ("num_entries_scanned_in_filter", C.c_int64),
                ("num_entries_scanned_post_filter", C.c_int64), ("num_total_docs", C.c_int64)]


class pg_result(C.Structure):
    _fields_ = [("stats", pg_stats), ("num_aggregations", C.c_int32), ("num_groups", C.c_int32),
                ("aggregations", C.POINTER(pg_agg_value)), ("group_ids", C.POINTER(C.c_int32)),
                ("group_aggregations", C.POINTER(pg_agg_value)), ("group_id_upper_bound", C.c_int32),
                ("num_groups_limit_reached", C.c_int32), ("device_ms", C.c_double), ("dominant_kernel_ms", C.c_double),
                ("profile_cycles", C.c_uint64 * 4), ("profile_waves", C.c_int32), ("dominant_kernel", C.c_int32),
                ("filter_entries_exact", C.c_int32), ("group_key_kind", C.c_int32), ("internal", C.c_void_p),
                ("group_ids64", C.POINTER(C.c_int64)), ("group_key_dict_ids", C.POINTER(C.c_int32))]


# every symbol include/pinot_gpu.h declares: (name, restype, argtypes)
_P = C.POINTER
ABI_SYMBOLS = [
    ("pg_init", C.c_int, [_P(pg_config)]),
    ("pg_shutdown", C.c_int, []),
    ("pg_last_error", C.c_char_p, []),
    ("pg_version", C.c_char_p, []),
    ("pg_device_info", C.c_int, [C.c_int32, C.c_char_p, C.c_int32, _P(C.c_int32), _P(C.c_uint64)]),
    ("pg_device_count", C.c_int, [_P(C.c_int32), _P(C.c_int32)]),
    ("pg_segment_open", C.c_int, [_P(pg_segment_desc), _P(C.c_void_p)]),
    ("pg_segment_close", C.c_int, [C.c_void_p]),
    ("pg_segment_num_docs", C.c_int, [C.c_void_p, _P(C.c_int32)]),
    ("pg_measure_stream_read", C.c_int, [C.c_int32, C.c_uint64, C.c_int32, _P(C.c_double)]),
    ("pg_segment_device_bytes", C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    ("pg_segment_plane_bytes", C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    ("pg_set_plane_budget", C.c_int, [C.c_uint64, _P(C.c_uint64)]),
    ("pg_query_check", C.c_int, [C.c_void_p, _P(pg_query)]),
    ("pg_execute", C.c_int, [C.c_void_p, _P(pg_query), _P(pg_result)]),
    ("pg_result_free", None, [_P(pg_result)]),
    ("pg_group_key_info", C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int64), _P(C.c_int32), _P(C.c_int32)]),
    ("pg_group_key_values", C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int64), C.c_int32, _P(C.c_int32)]),
    ("pg_execute_batch", C.c_int, [_P(C.c_void_p), _P(_P(pg_query)), C.c_int32, _P(pg_result), _P(C.c_int)]),
    ("pg_filter_bitmap", C.c_int, [C.c_void_p, _P(pg_query), _P(C.c_uint64), C.c_int64, _P(C.c_int64)]),
    ("pg_read_dict_ids", C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int32), C.c_int32, _P(C.c_int32)]),
    ("pg_read_int_values", C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int32), C.c_int32, _P(C.c_int32)]),
    ("pg_read_double_values", C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int32), C.c_int32, _P(C.c_double)]),
    ("pg_read_long_values", C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int32), C.c_int32, _P(C.c_int64)]),
]

CSRC_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
GPU_LIB_PATH = os.path.join(CSRC_DIR, "libpinot_gpu.so")
HOST_LIB_PATH = os.path.join(CSRC_DIR, "libpinot_host.so")


class PinotGpuError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("pinot_gpu status %d: %s" % (status, message))
        self.status = status


_gpu_lib = None


def load_gpu_library(path=None):
    """dlopen the in-tree libpinot_gpu.so and bind every ABI symbol.  Raises if it is missing: there is no fallback."""
    global _gpu_lib
    if _gpu_lib is not None and path is None:
        return _gpu_lib
    path = path or os.environ.get("PINOT_GPU_LIB") or GPU_LIB_PATH      # PINOT_GPU_LIB: A/B builds of the same ABI (tools/)
    if not os.path.exists(path):
        raise ImportError("HIP extension %s has not been built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C pinot_amd/csrc).  The engine has no CPU fallback." % path)
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, restype, argtypes in ABI_SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if path == (os.environ.get("PINOT_GPU_LIB") or GPU_LIB_PATH):
        _gpu_lib = lib
    return lib


def check(lib, status):
    if status != PG_OK:
        raise PinotGpuError(status, (lib.pg_last_error() or b"").decode("utf-8", "replace"))
