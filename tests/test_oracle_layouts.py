"""CPU tests: the oracle's layout restatements against known-answer bytes, independent numpy decoders, and the
product's C++ writers (pinot_amd/csrc/host/segment_writer.cpp) byte for byte."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import segment as S


def numpy_unpack(buf, bits, n):
    """Independent decoder: the file is one big-endian MSB-first bit string."""
    bitstream = np.unpackbits(np.asarray(buf, dtype=np.uint8))
    vals = bitstream[: n * bits].reshape(n, bits).astype(np.int64)
    weights = (1 << np.arange(bits - 1, -1, -1)).astype(np.int64)
    return (vals * weights).sum(axis=1).astype(np.int32)


def test_num_bits_per_value_examples():
    # PinotDataBitSet.java:44-52 javadoc examples + PinotDataBitSetTest.java:35-98 shift-count oracle
    lib = oracle.load()
    host = S.load_host_library()
    for max_value, want in [(0, 1), (1, 1), (2, 2), (9, 4), (113, 7), (255, 8), (256, 9), (65535, 16), (65536, 17),
                            (99999, 17), (999, 10), (2 ** 31 - 1, 31)]:
        assert lib.po_num_bits_per_value(max_value) == want
        assert host.ph_num_bits_per_value(max_value) == want
    for max_value in list(range(0, 5000)) + [2 ** k + d for k in range(12, 31) for d in (-1, 0, 1)]:
        want, v = 0, max_value
        while v > 0:
            v >>= 1
            want += 1
        want = max(want, 1)
        assert lib.po_num_bits_per_value(max_value) == want
        assert host.ph_num_bits_per_value(max_value) == want


def test_fixed_bit_known_answer_bytes():
    # Derived from PinotDataBitSet.writeInt (PinotDataBitSet.java:143-170); SURVEY.md appendix A.1 table.
    assert bytes(oracle.fixedbit_write([5, 2, 7, 1, 0], 3)) == bytes.fromhex("ab90")
    assert bytes(oracle.fixedbit_write([1, 65536, 131071], 17)) == bytes.fromhex("0000c0003fffe0")
    assert bytes(oracle.fixedbit_write([1, 0, 1, 1, 0, 0, 0, 1, 1], 1)) == bytes.fromhex("b180")
    assert bytes(oracle.fixedbit_write([0x7F, 0x01], 7)) == bytes.fromhex("fe04")
    assert bytes(oracle.fixedbit_write([0xAB, 0xCD], 8)) == bytes.fromhex("abcd")
    assert bytes(oracle.fixedbit_write([0x1FF, 0x001], 9)) == bytes.fromhex("ff8040")
    assert bytes(oracle.fixedbit_write([0x7FFFFFFF, 1], 31)) == bytes.fromhex("fffffffe00000004")


@pytest.mark.parametrize("bits", list(range(1, 32)))
def test_fixed_bit_roundtrip_all_widths(bits):
    # FixedBitIntReaderTest.java:52-84 (all 31 widths) / FixedBitSVForwardIndexReaderV2Test.java:75-110
    rng = np.random.default_rng(bits)
    n = 2049 + bits
    ids = rng.integers(0, 2 ** bits, n, dtype=np.int64).astype(np.int32)
    ids[:3] = [(1 << bits) - 1, 0, (1 << bits) - 1]
    buf = oracle.fixedbit_write(ids, bits)
    assert buf.shape[0] == (n * bits + 7) // 8
    assert (numpy_unpack(buf, bits, n) == ids).all()
    # product writer == oracle writer
    host = S.load_host_library()
    out = np.zeros(buf.shape[0], dtype=np.uint8)
    host.ph_fixedbit_pack(S._i32p(ids), n, bits, S._u8p(out), 3)
    assert (out == buf).all()
    # reader restatement: sequential (bulk read32 path), sparse, and the last docs (bounded read path)
    for doc_ids in (np.arange(n), np.arange(5, 5 + 200), np.arange(0, n, 7), np.array([n - 3, n - 2, n - 1]), np.array([n - 1])):
        got = oracle.read_dict_ids(buf, bits, n, doc_ids.astype(np.int32))
        assert (got == ids[doc_ids]).all()


def test_dictionary_search_and_range_lowering():
    values = np.array([-50, -3, 0, 7, 8, 100, 2 ** 31 - 1], dtype=np.int32)
    d = oracle.dict_write(values)
    assert bytes(d[:4]) == (-50).to_bytes(4, "big", signed=True)
    lib = oracle.load()
    for i, v in enumerate(values):
        assert lib.po_dict_insertion_index_of_int(oracle._u8p(d), len(values), int(v)) == i
        assert lib.po_dict_get_int(oracle._u8p(d), i) == int(v)
    # absent values return -(insertionPoint + 1), BaseImmutableDictionary.java:124-140
    assert lib.po_dict_insertion_index_of_int(oracle._u8p(d), len(values), 5) == -4
    assert lib.po_dict_insertion_index_of_int(oracle._u8p(d), len(values), -100) == -1
    assert oracle.index_of(d, len(values), 5) == -1
    # RangePredicateEvaluatorFactory.java:134-161
    assert oracle.lower_range(d, 7, lower=0, upper=8) == (2, 5)
    assert oracle.lower_range(d, 7, lower=0, upper=8, lower_inclusive=False, upper_inclusive=False) == (3, 4)
    assert oracle.lower_range(d, 7, lower=1, upper=9) == (3, 5)
    assert oracle.lower_range(d, 7, lower=None, upper=-51) == (0, 0)
    assert oracle.lower_range(d, 7, lower=101, upper=None) == (6, 7)
    assert oracle.lower_range(d, 7) == (0, 7)
    assert oracle.lower_range(d, 7, lower=9, upper=8) == (5, 5)


def test_raw_chunk_header_layout():
    # BaseChunkForwardIndexWriter.java:130-163 (version 2, 4-byte chunk offsets, PASS_THROUGH = 0)
    values = np.arange(2500, dtype=np.int32) * 3 - 7
    buf = oracle.raw_write(values, docs_per_chunk=1000)
    be = lambda off: int.from_bytes(bytes(buf[off:off + 4]), "big", signed=True)
    assert [be(o) for o in range(0, 28, 4)] == [2, 3, 1000, 4, 2500, 0, 28]
    header = 28 + 3 * 4
    assert [be(28 + 4 * c) for c in range(3)] == [header, header + 4000, header + 8000]
    assert buf.shape[0] == header + 2500 * 4
    assert be(header) == -7 and be(header + 4 * 2499) == 2499 * 3 - 7
    col = S.Column.raw("r", values, docs_per_chunk=1000)
    assert (col.fwd == buf).all()
    seg = S.SegmentData("t", 2500, [col])
    got = oracle.read_int_values(seg, 0, np.array([0, 1, 999, 1000, 2499], dtype=np.int32))
    assert (got == values[[0, 1, 999, 1000, 2499]]).all()


def _doc_words(doc_ids, num_words):
    w = np.zeros(num_words, dtype=np.uint64)
    for d in doc_ids:
        w[d >> 6] |= np.uint64(1) << np.uint64(d & 63)
    return w


@pytest.mark.parametrize("run_optimize", [False, True])
def test_roaring_serialization_all_container_kinds(run_optimize):
    rng = np.random.default_rng(5)
    sparse = np.sort(rng.choice(65536, 300, replace=False))                       # array container, key 0
    dense = 65536 + np.sort(rng.choice(65536, 30000, replace=False))             # bitset container, key 1
    runs = 3 * 65536 + np.concatenate([np.arange(10, 5000), np.arange(6000, 6003), [65535]])  # run container, key 3
    full = 4 * 65536 + np.arange(65536)                                          # full chunk, key 4
    doc_ids = np.concatenate([sparse, dense, runs, full]).astype(np.int32)
    lib = oracle.load()
    size = lib.po_roaring_serialize(oracle._i32p(doc_ids), len(doc_ids), int(run_optimize), None)
    out = np.zeros(size, dtype=np.uint8)
    assert lib.po_roaring_serialize(oracle._i32p(doc_ids), len(doc_ids), int(run_optimize), oracle._u8p(out)) == size
    cookie = int.from_bytes(bytes(out[:4]), "little")
    if run_optimize:
        assert cookie & 0xFFFF == 12347 and (cookie >> 16) + 1 == 4
        assert out[4] == 0b1100  # containers 2 and 3 (keys 3, 4) are run containers
    else:
        assert cookie == 12346 and int.from_bytes(bytes(out[4:8]), "little") == 4
    words, card = oracle.roaring_to_words(out, 5 * 1024)
    assert card == len(doc_ids)
    assert (words == _doc_words(doc_ids, 5 * 1024)).all()
    # product serializer agrees byte for byte
    host = S.load_host_library()
    out2 = np.zeros(size, dtype=np.uint8)
    assert host.ph_roaring_serialize(S._i32p(doc_ids), len(doc_ids), int(run_optimize), S._u8p(out2)) == size
    assert (out2 == out).all()


def test_roaring_hand_built_containers():
    # hand-assembled stream per the RoaringFormatSpec: cookie 12346, 1 container, key 2, array {1, 5, 65535}
    stream = (12346).to_bytes(4, "little") + (1).to_bytes(4, "little") + (2).to_bytes(2, "little") + (2).to_bytes(2, "little")
    stream += (16).to_bytes(4, "little") + b"".join(v.to_bytes(2, "little") for v in (1, 5, 65535))
    words, card = oracle.roaring_to_words(np.frombuffer(stream, dtype=np.uint8), 3 * 1024)
    assert card == 3
    assert (words == _doc_words([2 * 65536 + 1, 2 * 65536 + 5, 2 * 65536 + 65535], 3 * 1024)).all()
    # run cookie, 1 container (no offset header because n < 4): key 0, runs [3..6], [100..100]
    stream = (12347 | (0 << 16)).to_bytes(4, "little") + bytes([1]) + (0).to_bytes(2, "little") + (4).to_bytes(2, "little")
    stream += (2).to_bytes(2, "little") + (3).to_bytes(2, "little") + (3).to_bytes(2, "little") + (100).to_bytes(2, "little") + (0).to_bytes(2, "little")
    words, card = oracle.roaring_to_words(np.frombuffer(stream, dtype=np.uint8), 1024)
    assert card == 5
    assert (words == _doc_words([3, 4, 5, 6, 100], 1024)).all()


def test_inverted_index_layout():
    # BitmapInvertedIndexWriter.java:35-50: (C + 1) big-endian offsets (absolute), then the bitmaps
    rng = np.random.default_rng(2)
    n, card = 200000, 5
    ids = rng.integers(0, card, n).astype(np.int32)
    inv = oracle.inverted_build(ids, card)
    first = int.from_bytes(bytes(inv[:4]), "big")
    assert first == (card + 1) * 4
    assert int.from_bytes(bytes(inv[card * 4:card * 4 + 4]), "big") == inv.shape[0]
    nw = (n + 63) // 64 + 1024
    total = 0
    for d in range(card):
        o0 = int.from_bytes(bytes(inv[4 * d:4 * d + 4]), "big")
        o1 = int.from_bytes(bytes(inv[4 * d + 4:4 * d + 8]), "big")
        words, c = oracle.roaring_to_words(inv[o0:o1], nw)
        assert (words[: (n + 63) // 64] == _doc_words(np.nonzero(ids == d)[0], nw)[: (n + 63) // 64]).all()
        total += c
    assert total == n
    col = S.Column.from_dict_ids("c", np.arange(card, dtype=np.int32), ids, with_inverted=True)
    assert (col.inverted == inv).all()


def test_synthetic_generator_is_counter_based():
    a = S.synthetic_dict_ids(7, 0, 10000, 1000)
    b = S.synthetic_dict_ids(7, 2500, 100, 1000)
    assert (a[2500:2600] == b).all()
    assert a.min() >= 0 and a.max() < 1000 and len(np.unique(a)) > 990
    col = S.Column.synthetic_uniform("v", 10000, np.arange(1000, dtype=np.int32), 7)
    assert (numpy_unpack(col.fwd, col.bits, 10000) == a).all()


def test_bytes_written_by_the_reference_java_writers():
    """tests/golden/pinot_v1_segment_paddingOld.json holds the forward-index and dictionary files of a 5-doc segment that
    the reference's own Java writers produced (pinot-core/src/test/resources/data/paddingOld.tar.gz).  Every column has
    cardinality == totalDocs, so the decoded dictIds must be a permutation of 0..4 -- and re-encoding them with the
    oracle's and the product's writers must reproduce the reference's bytes exactly."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pinot_v1_segment_paddingOld.json")))
    n = g["total_docs"]
    assert n == 5
    host = S.load_host_library()
    expect = {"age": [4, 2, 3, 0, 1], "percent": [0, 2, 4, 3, 1], "outgoingName1": [4, 3, 1, 0, 2], "name": [1, 0, 0, 0, 1]}
    for name, col in g["columns"].items():
        fwd = np.frombuffer(bytes.fromhex(col["fwd_hex"]), dtype=np.uint8)
        bits, card = col["bitsPerElement"], col["cardinality"]
        assert oracle.load().po_num_bits_per_value(card - 1) == bits == host.ph_num_bits_per_value(card - 1)
        assert fwd.shape[0] == (n * bits + 7) // 8
        ids = oracle.read_dict_ids(fwd, bits, n, np.arange(n, dtype=np.int32))
        assert ids.tolist() == expect[name]
        if card == n:
            assert sorted(ids.tolist()) == list(range(n))
        assert bytes(oracle.fixedbit_write(ids, bits)) == bytes(fwd)            # oracle writer == Java writer
        out = np.zeros(fwd.shape[0], dtype=np.uint8)
        host.ph_fixedbit_pack(S._i32p(ids), n, bits, S._u8p(out), 1)
        assert bytes(out) == bytes(fwd)                                          # product writer == Java writer
    age = g["columns"]["age"]
    d = np.frombuffer(bytes.fromhex(age["dict_hex"]), dtype=np.uint8)
    values = [oracle.load().po_dict_get_int(oracle._u8p(d), i) for i in range(5)]
    assert values == [617, 824, 837, 1209, 1228] and values == sorted(values)
    assert bytes(oracle.dict_write(np.array(values, dtype=np.int32))) == bytes(d)   # dictionary writer == Java writer
    assert oracle.index_of(d, 5, 1209) == 3
