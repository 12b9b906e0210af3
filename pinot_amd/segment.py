"""Host-side segment construction for tests and benchmarks (Python plumbing over libpinot_host.so).

Column buffers are produced by the product's C++ writers (pinot_amd/csrc/host/segment_writer.cpp) in the
reference's on-disk layouts and handed to the engine through `pg_segment_desc`, exactly what a JNI shim
would do with the `PinotDataBuffer`s of an `ImmutableSegment`.
"""
import ctypes as C
import os

import numpy as np

from . import _abi

_host_lib = None


def load_host_library():
    global _host_lib
    if _host_lib is not None:
        return _host_lib
    if not os.path.exists(_abi.HOST_LIB_PATH):
        raise ImportError("%s has not been built (make -C pinot_amd/csrc)" % _abi.HOST_LIB_PATH)
    lib = C.CDLL(_abi.HOST_LIB_PATH)
    i32p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    lib.ph_num_bits_per_value.restype = C.c_int32
    lib.ph_num_bits_per_value.argtypes = [C.c_int32]
    lib.ph_fixedbit_size.restype = C.c_int64
    lib.ph_fixedbit_size.argtypes = [C.c_int64, C.c_int32]
    lib.ph_fixedbit_pack.restype = None
    lib.ph_fixedbit_pack.argtypes = [i32p, C.c_int64, C.c_int32, u8p, C.c_int32]
    lib.ph_generate_packed_uniform.restype = None
    lib.ph_generate_packed_uniform.argtypes = [C.c_uint64, C.c_int64, C.c_int32, C.c_int32, u8p, C.c_int32]
    lib.ph_generate_uniform.restype = None
    lib.ph_generate_uniform.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int32, i32p]
    lib.ph_dict_write_int.restype = None
    lib.ph_dict_write_int.argtypes = [i32p, C.c_int32, u8p]
    lib.ph_raw_size_v2.restype = C.c_int64
    lib.ph_raw_size_v2.argtypes = [C.c_int32, C.c_int32]
    lib.ph_raw_write_int_v2.restype = None
    lib.ph_raw_write_int_v2.argtypes = [i32p, C.c_int32, C.c_int32, u8p]
    lib.ph_dict_write_fixed.restype = None
    lib.ph_dict_write_fixed.argtypes = [C.c_void_p, C.c_int32, C.c_int32, u8p]
    lib.ph_raw_size_fixed_v2.restype = C.c_int64
    lib.ph_raw_size_fixed_v2.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.ph_raw_write_fixed_v2.restype = None
    lib.ph_raw_write_fixed_v2.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, u8p]
    lib.ph_roaring_serialize.restype = C.c_int64
    lib.ph_roaring_serialize.argtypes = [i32p, C.c_int64, C.c_int32, u8p]
    lib.ph_inverted_build.restype = C.c_int64
    lib.ph_inverted_build.argtypes = [i32p, C.c_int32, C.c_int32, C.c_int32, u8p]
    _host_lib = lib
    return lib


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def host_threads():
    return max(1, min(32, os.cpu_count() or 1))


# stored type -> numpy host dtype of the values
TYPE_DTYPES = {_abi.PG_TYPE_INT: np.int32, _abi.PG_TYPE_LONG: np.int64, _abi.PG_TYPE_FLOAT: np.float32, _abi.PG_TYPE_DOUBLE: np.float64}


def stored_type_of(dtype):
    dtype = np.dtype(dtype)
    for t, d in TYPE_DTYPES.items():
        if np.dtype(d) == dtype:
            return t
    raise TypeError("no Pinot stored type for %s" % dtype)


def roaring_serialize(doc_ids, num_docs=None, run_optimize=True):
    """One RoaringBitmap of ascending docIds in the portable serialization (what RoaringBitmap.serialize writes)."""
    lib = load_host_library()
    ids = np.ascontiguousarray(doc_ids, dtype=np.int32)
    lib.ph_roaring_serialize.restype = C.c_int64
    lib.ph_roaring_serialize.argtypes = [C.POINTER(C.c_int32), C.c_int64, C.c_int32, C.POINTER(C.c_uint8)]
    size = int(lib.ph_roaring_serialize(_i32p(ids), int(ids.shape[0]), int(run_optimize), None))
    out = np.zeros(size, dtype=np.uint8)
    lib.ph_roaring_serialize(_i32p(ids), int(ids.shape[0]), int(run_optimize), _u8p(out))
    return out


class Column:
    """One single-value numeric column: forward index (+ dictionary, + optional inverted index) as raw bytes."""

    def __init__(self, name, encoding, bits, cardinality, fwd, dictionary=None, inverted=None, dict_values=None,
                 stored_type=_abi.PG_TYPE_INT):
        self.stored_type = stored_type
        self.name = name
        self.encoding = encoding
        self.bits = bits
        self.cardinality = cardinality
        self.fwd = fwd                  # np.uint8
        self.dictionary = dictionary    # np.uint8 big-endian int32s
        self.inverted = inverted        # np.uint8 or None
        self.dict_values = dict_values  # np.int32 sorted values (host convenience, not handed to the engine)
        self.null_vector = None         # np.uint8: the <column>.bitmap.nullvalue file (one serialized RoaringBitmap) or None

    def with_nulls(self, null_mask):
        """Attach a null value vector (NullValueVectorCreator: a RoaringBitmap of the null docIds; no file when there are none).
        The caller has already stored the default null value at those docs, like the segment creator does."""
        null_mask = np.ascontiguousarray(null_mask, dtype=bool)
        if null_mask.any():
            self.null_vector = roaring_serialize(np.flatnonzero(null_mask).astype(np.int32), len(null_mask))
        return self

    @staticmethod
    def dict_encoded(name, values, with_inverted=False, run_optimize=True):
        """Dictionary-encode `values` the way SegmentDictionaryCreator + FixedBitSVForwardIndexWriter do."""
        lib = load_host_library()
        values = np.ascontiguousarray(values, dtype=np.int32)
        dict_values, dict_ids = np.unique(values, return_inverse=True)
        dict_ids = np.ascontiguousarray(dict_ids, dtype=np.int32)
        return Column.from_dict_ids(name, dict_values.astype(np.int32), dict_ids, with_inverted, run_optimize)

    @staticmethod
    def from_dict_ids(name, dict_values, dict_ids, with_inverted=False, run_optimize=True):
        lib = load_host_library()
        dict_values = np.ascontiguousarray(dict_values, dtype=np.int32)
        dict_ids = np.ascontiguousarray(dict_ids, dtype=np.int32)
        card = int(dict_values.shape[0])
        n = int(dict_ids.shape[0])
        bits = int(lib.ph_num_bits_per_value(card - 1))
        fwd = np.zeros(int(lib.ph_fixedbit_size(n, bits)), dtype=np.uint8)
        if n:
            lib.ph_fixedbit_pack(_i32p(dict_ids), n, bits, _u8p(fwd), host_threads())
        dictionary = np.zeros(card * 4, dtype=np.uint8)
        lib.ph_dict_write_int(_i32p(dict_values), card, _u8p(dictionary))
        inverted = None
        if with_inverted:
            size = int(lib.ph_inverted_build(_i32p(dict_ids), n, card, int(run_optimize), None))
            inverted = np.zeros(size, dtype=np.uint8)
            lib.ph_inverted_build(_i32p(dict_ids), n, card, int(run_optimize), _u8p(inverted))
        return Column(name, _abi.PG_FWD_FIXED_BIT_DICT, bits, card, fwd, dictionary, inverted, dict_values)

    @staticmethod
    def dict_encoded_typed(name, values, with_inverted=False):
        """LONG / FLOAT / DOUBLE (or INT) dictionary column; the stored type follows the numpy dtype of `values`."""
        lib = load_host_library()
        values = np.ascontiguousarray(values)
        st = stored_type_of(values.dtype)
        if st == _abi.PG_TYPE_INT:
            return Column.dict_encoded(name, values, with_inverted)
        dict_values, dict_ids = np.unique(values, return_inverse=True)     # ascending, like SegmentDictionaryCreator
        dict_values = np.ascontiguousarray(dict_values, dtype=values.dtype)
        dict_ids = np.ascontiguousarray(dict_ids, dtype=np.int32)
        card, n = int(dict_values.shape[0]), int(dict_ids.shape[0])
        bits = int(lib.ph_num_bits_per_value(card - 1))
        fwd = np.zeros(int(lib.ph_fixedbit_size(n, bits)), dtype=np.uint8)
        if n:
            lib.ph_fixedbit_pack(_i32p(dict_ids), n, bits, _u8p(fwd), host_threads())
        dictionary = np.zeros(card * values.dtype.itemsize, dtype=np.uint8)
        lib.ph_dict_write_fixed(dict_values.ctypes.data, card, values.dtype.itemsize, _u8p(dictionary))
        inverted = None
        if with_inverted:
            size = int(lib.ph_inverted_build(_i32p(dict_ids), n, card, 1, None))
            inverted = np.zeros(size, dtype=np.uint8)
            lib.ph_inverted_build(_i32p(dict_ids), n, card, 1, _u8p(inverted))
        return Column(name, _abi.PG_FWD_FIXED_BIT_DICT, bits, card, fwd, dictionary, inverted, dict_values, stored_type=st)

    @staticmethod
    def raw_typed(name, values, docs_per_chunk=1000):
        """Raw PASS_THROUGH v2 chunks of INT / LONG / FLOAT / DOUBLE values."""
        lib = load_host_library()
        values = np.ascontiguousarray(values)
        st = stored_type_of(values.dtype)
        n, es = int(values.shape[0]), values.dtype.itemsize
        fwd = np.zeros(int(lib.ph_raw_size_fixed_v2(n, docs_per_chunk, es)), dtype=np.uint8)
        lib.ph_raw_write_fixed_v2(values.ctypes.data, n, docs_per_chunk, es, _u8p(fwd))
        return Column(name, _abi.PG_FWD_RAW_FIXED_BYTE, 8 * es, 0, fwd, stored_type=st)

    @staticmethod
    def synthetic_uniform(name, num_docs, dict_values, seed):
        """dictId_i = splitmix64-based uniform over the dictionary, generated and packed by the C++ writer."""
        lib = load_host_library()
        dict_values = np.ascontiguousarray(dict_values, dtype=np.int32)
        card = int(dict_values.shape[0])
        bits = int(lib.ph_num_bits_per_value(card - 1))
        fwd = np.zeros(int(lib.ph_fixedbit_size(num_docs, bits)), dtype=np.uint8)
        lib.ph_generate_packed_uniform(seed, num_docs, card, bits, _u8p(fwd), host_threads())
        dictionary = np.zeros(card * 4, dtype=np.uint8)
        lib.ph_dict_write_int(_i32p(dict_values), card, _u8p(dictionary))
        return Column(name, _abi.PG_FWD_FIXED_BIT_DICT, bits, card, fwd, dictionary, None, dict_values)

    @staticmethod
    def raw(name, values, docs_per_chunk=1000):
        lib = load_host_library()
        values = np.ascontiguousarray(values, dtype=np.int32)
        n = int(values.shape[0])
        fwd = np.zeros(int(lib.ph_raw_size_v2(n, docs_per_chunk)), dtype=np.uint8)
        lib.ph_raw_write_int_v2(_i32p(values), n, docs_per_chunk, _u8p(fwd))
        return Column(name, _abi.PG_FWD_RAW_FIXED_BYTE, 32, 0, fwd)

    def value_of(self, dict_id):
        v = self.dict_values[dict_id]
        return int(v) if self.stored_type in (_abi.PG_TYPE_INT, _abi.PG_TYPE_LONG) else float(v)


def synthetic_dict_ids(seed, start, count, cardinality):
    lib = load_host_library()
    out = np.zeros(count, dtype=np.int32)
    lib.ph_generate_uniform(seed, start, count, cardinality, _i32p(out))
    return out


class SegmentData:
    """Host buffers of one segment + the `pg_segment_desc` that points at them."""

    def __init__(self, name, num_docs, columns, device_id=-1):
        self.name = name
        self.num_docs = int(num_docs)
        self.columns = list(columns)
        self._name_b = name.encode()
        self._col_names = [c.name.encode() for c in self.columns]
        self._descs = (_abi.pg_column_desc * max(len(self.columns), 1))()
        for i, c in enumerate(self.columns):
            d = self._descs[i]
            d.name = self._col_names[i]
            d.stored_type = c.stored_type
            d.fwd_encoding = c.encoding
            d.bits_per_value = c.bits
            d.cardinality = c.cardinality
            d.fwd_data = c.fwd.ctypes.data
            d.fwd_size = c.fwd.nbytes
            if c.dictionary is not None:
                d.dict_data = c.dictionary.ctypes.data
                d.dict_size = c.dictionary.nbytes
            if c.inverted is not None:
                d.inv_data = c.inverted.ctypes.data
                d.inv_size = c.inverted.nbytes
            if c.null_vector is not None:
                d.null_data = c.null_vector.ctypes.data
                d.null_size = c.null_vector.nbytes
        self.desc = _abi.pg_segment_desc()
        self.desc.name = self._name_b
        self.desc.crc = 0
        self.desc.num_docs = self.num_docs
        self.desc.num_columns = len(self.columns)
        self.desc.columns = self._descs
        self.desc.device_id = device_id

    def column_index(self, name):
        for i, c in enumerate(self.columns):
            if c.name == name:
                return i
        raise KeyError(name)

    def column(self, name):
        return self.columns[self.column_index(name)]
