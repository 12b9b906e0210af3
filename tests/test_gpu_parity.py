"""GPU parity tests: the HIP path behind the C ABI against the CPU oracle on the same seeded inputs (bit exact)."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

pytestmark = pytest.mark.gpu

ALL_AGGS = lambda c: [(Q.COUNT, -1), (Q.SUM, c), (Q.MIN, c), (Q.MAX, c), (Q.AVG, c)]


def run_both(engine, seg, spec, check_stats=True):
    with engine.open(seg) as gseg:
        got = gseg.execute(spec)
    want = oracle.execute(seg, spec)
    H.assert_results_equal(got, want, check_stats)
    return got, want


def test_library_is_the_hip_build(engine):
    name, cus, hbm = engine.device_info()
    assert name.startswith("gfx950"), name
    assert cus >= 200 and hbm > 200 * 2 ** 30


@pytest.mark.parametrize("bits", list(range(1, 23)))
def test_every_bit_width_filter_and_aggregate(engine, bits):
    """FixedBitIntReaderTest.java:52-84 runs all 31 widths; widths whose dictionary fits comfortably use
    full-range dictIds here (cardinality = 2^bits or one less)."""
    rng = np.random.default_rng(100 + bits)
    card = 2 ** bits - (bits % 2) if bits > 1 else 2
    n = 4099 + 13 * bits
    col, ids, dv = H.random_dict_column(rng, "v", n, card, value_stride=5)
    seg = S.SegmentData("w%d" % bits, n, [col])
    lo, hi = card // 4, max(card // 4 + 1, (3 * card) // 4)
    spec = Q.QuerySpec(ALL_AGGS(0), filter=Q.leaf(Q.Pred.dict_range(0, lo, hi)))
    got, _ = run_both(engine, seg, spec)
    m = (ids >= lo) & (ids < hi)
    assert got.aggregations[0].count == int(m.sum())
    assert got.aggregations[1].sum_i64 == int(dv[ids[m]].astype(np.int64).sum())
    run_both(engine, seg, Q.QuerySpec(ALL_AGGS(0)))


@pytest.mark.parametrize("bits", [23, 25, 26, 29, 31])
def test_wide_bit_widths_with_forced_width(engine, bits):
    """Widths whose real dictionary would be GBs: the stream is packed at `bits` bits with a small dictionary, so
    the kernel's wide decode path (26..31 bits) is exercised on values whose low bits vary."""
    rng = np.random.default_rng(bits)
    card, n = 50000, 10007
    dict_values = (np.arange(card, dtype=np.int64) * 11 - 70000).astype(np.int32)
    ids = rng.integers(0, card, n).astype(np.int32)
    col = S.Column.from_dict_ids("v", dict_values, ids)
    host = S.load_host_library()
    col.bits = bits
    col.fwd = np.zeros(int(host.ph_fixedbit_size(n, bits)), dtype=np.uint8)
    host.ph_fixedbit_pack(S._i32p(ids), n, bits, S._u8p(col.fwd), 2)
    seg = S.SegmentData("wide%d" % bits, n, [col])
    run_both(engine, seg, Q.QuerySpec(ALL_AGGS(0), filter=Q.leaf(Q.Pred.dict_range(0, 1000, 40000))))


def test_full_range_26_bit_dictionary(engine):
    """A real 2^26-entry dictionary (256 MB) so that dictIds use all 26 bits (kWide path, 5-byte spans)."""
    rng = np.random.default_rng(26)
    bits, n = 26, 300000
    card = 2 ** bits
    dict_values = np.arange(card, dtype=np.int32) - 2 ** 25
    ids = rng.integers(0, card, n).astype(np.int32)
    ids[:4] = [card - 1, 0, card - 1, 1]
    col = S.Column.from_dict_ids("v", dict_values, ids)
    assert col.bits == 26
    seg = S.SegmentData("full26", n, [col])
    got, _ = run_both(engine, seg, Q.QuerySpec(ALL_AGGS(0), filter=Q.leaf(Q.Pred.dict_range(0, card // 3, card))))
    m = ids >= card // 3
    assert got.aggregations[1].sum_i64 == int(dict_values[ids[m]].astype(np.int64).sum())


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 2047, 2048, 2049, 4096, 10000])
def test_ragged_sizes(engine, n):
    rng = np.random.default_rng(n)
    card = 37
    dict_values = np.arange(card, dtype=np.int32) * 3 + 1
    ids = rng.integers(0, card, n).astype(np.int32)
    seg = S.SegmentData("n%d" % n, n, [S.Column.from_dict_ids("v", dict_values, ids), S.Column.raw("r", (ids * 2 - 5).astype(np.int32))])
    run_both(engine, seg, Q.QuerySpec(ALL_AGGS(0) + [(Q.SUM, 1), (Q.MAX, 1)], filter=Q.leaf(Q.Pred.dict_range(0, 5, 30))))
    run_both(engine, seg, Q.QuerySpec(ALL_AGGS(0)))
    run_both(engine, seg, Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(0, 5, 30)), group_by=[0]))


def _three_column_segment(rng, n, with_inverted=False):
    a, ida, dva = H.random_dict_column(rng, "a", n, 1000, with_inverted=with_inverted)
    b, idb, dvb = H.random_dict_column(rng, "b", n, 13, with_inverted=with_inverted)
    c, idc, dvc = H.random_dict_column(rng, "c", n, 70000, value_stride=3)
    raw_vals = rng.integers(-1000, 1000, n).astype(np.int32)
    seg = S.SegmentData("three", n, [a, b, c, S.Column.raw("r", raw_vals)])
    return seg, (ida, idb, idc, raw_vals)


def test_filter_trees_and_predicate_kinds(engine):
    rng = np.random.default_rng(7)
    n = 50021
    seg, (ida, idb, idc, raw) = _three_column_segment(rng, n)
    P = Q.Pred
    filters = {
        "and": Q.and_(Q.leaf(P.dict_range(0, 100, 900)), Q.leaf(P.dict_range(1, 2, 9))),
        "or": Q.or_(Q.leaf(P.dict_range(0, 0, 10)), Q.leaf(P.dict_range(2, 60000, 70000))),
        "not": Q.not_(Q.leaf(P.dict_range(1, 3, 4))),
        "neq": Q.leaf(P.dict_range(1, 3, 4, exclusive=True)),
        "in": Q.leaf(P.dict_set(0, [1, 5, 33, 64, 999], 1000)),
        "not_in": Q.leaf(P.dict_set(1, [0, 12], 13, exclusive=True)),
        "raw": Q.leaf(P.raw_range(3, -10, 500)),
        "raw_and_dict": Q.and_(Q.leaf(P.raw_range(3, 0, 2 ** 31 - 1)), Q.leaf(P.dict_set(2, list(range(0, 70000, 3)), 70000))),
        "nested": Q.and_(Q.leaf(P.dict_range(0, 50, 950)),
                         Q.or_(Q.leaf(P.dict_range(1, 0, 3)), Q.not_(Q.leaf(P.dict_set(2, list(range(100, 30000)), 70000)))),
                         Q.leaf(P.match_all()), Q.not_(Q.leaf(P.match_none()))),
        "none": Q.and_(Q.leaf(P.dict_range(0, 0, 1000)), Q.leaf(P.match_none())),
        "empty_range": Q.leaf(P.dict_range(0, 10, 10)),
        "all": Q.leaf(P.match_all()),
        "five_leaves": Q.or_(Q.and_(Q.leaf(P.dict_range(0, 0, 500)), Q.leaf(P.dict_range(1, 0, 6)), Q.leaf(P.dict_range(2, 0, 35000))),
                             Q.and_(Q.leaf(P.dict_range(0, 500, 1000)), Q.leaf(P.raw_range(3, -5, 5)))),
    }
    aggs = [(Q.COUNT, -1), (Q.SUM, 2), (Q.MIN, 0), (Q.MAX, 3), (Q.AVG, 1), (Q.SUM, 3), (Q.MAX, 2)]
    with engine.open(seg) as gseg:
        for name, flt in filters.items():
            spec = Q.QuerySpec(aggs, filter=flt)
            H.assert_results_equal(gseg.execute(spec), oracle.execute(seg, spec))
            gw, gc = gseg.filter_bitmap(Q.QuerySpec([], filter=flt))
            ow, oc = oracle.filter_bitmap(seg, Q.QuerySpec([], filter=flt))
            assert gc == oc, name
            assert (gw == ow).all(), name


def test_same_column_filter_and_sum_like_c2a(engine):
    rng = np.random.default_rng(11)
    n = 300007
    v, ids, dv = H.random_dict_column(rng, "v", n, 100000, value_stride=7)
    assert v.bits == 17
    seg = S.SegmentData("c2a", n, [v])
    for lo, hi in ((45000, 55000), (25000, 75000), (5000, 95000)):
        got, _ = run_both(engine, seg, Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(0, lo, hi))))
        m = (ids >= lo) & (ids < hi)
        assert got.aggregations[0].sum_i64 == int(dv[ids[m]].astype(np.int64).sum())


def test_two_column_filtered_sum_like_c2b(engine):
    n = 1000003
    v = S.Column.synthetic_uniform("v", n, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=1)
    f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2)
    assert (v.bits, f.bits) == (17, 10)
    seg = S.SegmentData("c2b", n, [v, f])
    for t in (10, 100, 500):
        run_both(engine, seg, Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, t))))


def test_group_by_lds_table(engine):
    rng = np.random.default_rng(3)
    n = 200003
    k, idk, _ = H.random_dict_column(rng, "k", n, 1000)
    a, ida, dva = H.random_dict_column(rng, "a", n, 100000, value_stride=7)
    b, idb, dvb = H.random_dict_column(rng, "b", n, 65536, value_stride=2)
    f, idf, _ = H.random_dict_column(rng, "f", n, 1000)
    seg = S.SegmentData("c3", n, [k, a, b, f, S.Column.raw("r", rng.integers(-50, 50, n).astype(np.int32))])
    aggs = [(Q.SUM, 1), (Q.MAX, 2), (Q.COUNT, -1), (Q.MIN, 2), (Q.AVG, 1), (Q.SUM, 4), (Q.MIN, 4)]  # 5 distinct device accumulators
    got, _ = run_both(engine, seg, Q.QuerySpec(aggs, group_by=[0]))
    assert len(got.groups) == 1000
    g7 = ida[idk == 7]
    assert got.groups[7][0].sum_i64 == int(dva[g7].astype(np.int64).sum())
    run_both(engine, seg, Q.QuerySpec(aggs, filter=Q.leaf(Q.Pred.dict_range(3, 0, 100)), group_by=[0]))
    run_both(engine, seg, Q.QuerySpec(aggs, filter=Q.leaf(Q.Pred.dict_range(3, 0, 1)), group_by=[0]))
    run_both(engine, seg, Q.QuerySpec(aggs, filter=Q.leaf(Q.Pred.match_none()), group_by=[0]))


def test_group_by_multiple_keys_and_global_table(engine):
    rng = np.random.default_rng(4)
    n = 120001
    k1, id1, _ = H.random_dict_column(rng, "k1", n, 90)
    k2, id2, _ = H.random_dict_column(rng, "k2", n, 11)
    k3, id3, _ = H.random_dict_column(rng, "k3", n, 10)
    v, idv, dvv = H.random_dict_column(rng, "v", n, 5000)
    seg = S.SegmentData("mk", n, [k1, k2, k3, v])
    aggs = [(Q.COUNT, -1), (Q.SUM, 3), (Q.MAX, 3), (Q.MIN, 3)]
    got, _ = run_both(engine, seg, Q.QuerySpec(aggs, group_by=[0, 1]))              # 990 groups, LDS table
    gid = 5 + 3 * 90   # dictIds (5, 3): k2 * card(k1) + k1  (DictionaryBasedGroupKeyGenerator.java:312-317)
    m = (id1 == 5) & (id2 == 3)
    assert got.groups[gid][0].count == int(m.sum())
    got, _ = run_both(engine, seg, Q.QuerySpec(aggs, group_by=[0, 1, 2]))           # 9900 groups x 4 words: global-memory table
    assert got.group_id_upper_bound == 9900
    gid = 5 + 3 * 90 + 7 * 990
    m = (id1 == 5) & (id2 == 3) & (id3 == 7)
    assert (gid in got.groups) == bool(m.any())
    run_both(engine, seg, Q.QuerySpec(aggs, filter=Q.leaf(Q.Pred.dict_range(3, 100, 2000)), group_by=[2, 0]))
    got, _ = run_both(engine, seg, Q.QuerySpec(aggs, group_by=[0, 1, 3]))   # 90 * 11 * 5000 raw keys: the IntMapBasedHolder range, HBM table
    assert got.group_id_upper_bound == 90 * 11 * 5000 and len(got.groups) > 10000
    got, want = run_both(engine, seg, Q.QuerySpec(aggs, group_by=[3, 3, 0]))   # 5000 * 5000 * 90 raw keys are not an int: LongMapBasedHolder, a hashed HBM table
    assert got.group_key_kind == want.group_key_kind == 1 and got.group_keys == want.group_keys and got.group_ids64 == want.group_ids64


def test_group_by_whole_int_map_range(engine):
    """Cardinality products up to Integer.MAX_VALUE stay one direct-indexed HBM table (IntMapBasedHolder's range,
    DictionaryBasedGroupKeyGenerator.java:164-184): above 2^24 slots the key multiplies leave the 24-bit form."""
    rng = np.random.default_rng(41)
    n = 150001
    k1, id1, _ = H.random_dict_column(rng, "k1", n, 5000)
    k2, id2, _ = H.random_dict_column(rng, "k2", n, 11)
    v, idv, dvv = H.random_dict_column(rng, "v", n, 3000, value_stride=3)
    rl = S.Column.raw_typed("rl", rng.integers(-10 ** 12, 10 ** 12, n).astype(np.int64))
    seg = S.SegmentData("wide", n, [k1, k2, v, rl])
    aggs = [(Q.COUNT, -1), (Q.SUM, 2), (Q.MAX, 2)]
    got, _ = run_both(engine, seg, Q.QuerySpec(aggs, group_by=[0, 0]))               # 25 M slots, multiplier 5000
    assert got.group_id_upper_bound == 25_000_000 and set(got.groups) == set((id1.astype(np.int64) * 5001).tolist())
    got, _ = run_both(engine, seg, Q.QuerySpec(aggs, group_by=[0, 0, 1]))            # 275 M slots, third multiplier 25 M >= 2^24
    assert got.group_id_upper_bound == 275_000_000
    keys = id1.astype(np.int64) * 5001 + id2.astype(np.int64) * 25_000_000
    assert set(got.groups) == set(keys.tolist())
    big = int(keys.max())
    m = keys == big
    assert got.groups[big][0].count == int(m.sum()) and got.groups[big][1].sum_i64 == int(dvv[idv[m]].astype(np.int64).sum())
    run_both(engine, seg, Q.QuerySpec(aggs, filter=Q.leaf(Q.Pred.dict_range(2, 100, 900)), group_by=[0, 0, 1]))      # masked lane-private path
    # a raw LONG range leaf is not in the lane-private filter: the LDS-staged kernel aggregates the same key space
    run_both(engine, seg, Q.QuerySpec(aggs, filter=Q.leaf(Q.Pred.raw_range(3, -10 ** 11, 10 ** 11)), group_by=[0, 0, 1]))
    # numGroupsLimit below the groups that exist: the first-doc pass walks the 275 M slots
    got, want = run_both(engine, seg, Q.QuerySpec(aggs, group_by=[0, 0, 1], num_groups_limit=300))
    assert len(got.groups) == 300 and got.num_groups_limit_reached
    # a raw 8-byte aggregation input goes through group_typed_direct_kernel
    run_both(engine, seg, Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 3), (Q.MIN, 3)], group_by=[0, 0, 1]))


@pytest.mark.parametrize("n", [1, 2047, 4 * 2048, 4 * 2048 + 5, 9 * 2048 + 77, 300_007])
def test_narrow_column_filters_four_tiles_at_a_time(engine, n):
    """scan_narrow_kernel: COUNT(*) / the docId bitmap of filters over dictionary columns of 1..8 bits (any AND / OR / NOT tree of dictId
    ranges and dictId sets), four tiles per wave and iteration.  Whole quads, ragged quads, a single doc."""
    rng = np.random.default_rng(77 + n)
    cols, ids = [], []
    for b in range(1, 9):
        card = 2 ** b - (1 if b in (3, 6) else 0)
        c, i, _ = H.random_dict_column(rng, "c%d" % b, n, card)
        assert c.bits == b
        cols.append(c)
        ids.append(i)
    seg = S.SegmentData("narrow", n, cols)
    R = lambda col, lo, hi, ex=False: Q.leaf(Q.Pred.dict_range(col, lo, hi, exclusive=ex))
    trees = [R(3, 3, 4), R(0, 0, 1), R(7, 7, 100), R(5, 0, 33, ex=True),
             Q.and_(R(3, 3, 4), R(5, 5, 6), R(7, 7, 8)),
             Q.or_(R(1, 0, 2), Q.not_(R(4, 8, 20)), R(6, 100, 101)),
             Q.and_(Q.or_(R(2, 1, 3), R(3, 0, 9)), Q.not_(Q.and_(R(4, 0, 16), R(5, 10, 60, ex=True))), R(6, 0, 127)),
             Q.not_(Q.not_(Q.or_(Q.and_(R(0, 1, 2), R(1, 1, 3)), Q.and_(R(2, 2, 7), Q.leaf(Q.Pred.match_all())), Q.leaf(Q.Pred.match_none()))))]
    # IN / NOT IN over narrow columns (round 6b): a set of at most eight words, one register up to five bits, looked up in LDS above
    SET = lambda col, members, ex=False: Q.leaf(Q.Pred.dict_set(col, sorted(members), cols[col].cardinality, exclusive=ex))
    trees += [SET(0, [1]), SET(2, [0, 3, 6]), SET(4, [0, 7, 19, 31], ex=True), SET(5, [1, 2, 33, 62]), SET(6, range(3, 120, 7)), SET(7, [0, 31, 32, 63, 64, 200, 255], ex=True),
              Q.and_(SET(3, [1, 2, 9, 15]), R(5, 5, 40), SET(7, range(0, 256, 3))),
              Q.or_(Q.not_(SET(6, [5, 50, 100])), Q.and_(SET(1, [0, 3]), SET(4, range(0, 32, 2))), R(2, 6, 7))]
    with engine.open(seg) as g:
        for tree in trees:
            spec = Q.QuerySpec([(Q.COUNT, -1)], filter=tree)
            got = g.execute(spec)
            want = oracle.execute(seg, spec)
            H.assert_results_equal(got, want)
            assert got.dominant_kernel == "scan_narrow_kernel"
            words, card = g.filter_bitmap(spec)
            owords, ocard = oracle.filter_bitmap(seg, spec)
            assert card == ocard == got.aggregations[0].count and np.array_equal(words, owords)
        # five masks deep, a 9-bit column, an aggregated column: scan_private_kernel keeps those
        deep = Q.or_(R(0, 0, 1), Q.and_(R(1, 0, 2), Q.or_(R(2, 0, 3), Q.and_(R(3, 0, 4), Q.or_(R(4, 0, 5), R(5, 0, 6))))))
        got = g.execute(Q.QuerySpec([(Q.COUNT, -1)], filter=deep))
        H.assert_results_equal(got, oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1)], filter=deep)))
        assert got.dominant_kernel != "scan_narrow_kernel"
        got = g.execute(Q.QuerySpec([(Q.COUNT, -1), (Q.MAX, 7)], filter=R(3, 3, 4)))
        assert got.dominant_kernel != "scan_narrow_kernel"
    m = (ids[3] == 3) & (ids[5] == 5) & (ids[7] == 7)
    with engine.open(seg) as g:
        assert g.execute(Q.QuerySpec([(Q.COUNT, -1)], filter=trees[4])).aggregations[0].count == int(m.sum())


@pytest.mark.parametrize("run_optimize", [False, True])
def test_inverted_index_leaves(engine, run_optimize):
    """InvertedIndexFilterOperator + AndDocIdSet: postings (array / bitset / run containers) expanded on device,
    ANDed with each other and with scan leaves.  Config 5's shape at CI size."""
    rng = np.random.default_rng(9)
    n = 400009
    p, idp, _ = H.random_dict_column(rng, "p", n, 2, with_inverted=True, run_optimize=run_optimize)        # bitset containers
    q, idq, _ = H.random_dict_column(rng, "q", n, 64, with_inverted=True, run_optimize=run_optimize)       # array containers
    r, idr, _ = H.random_dict_column(rng, "r", n, 40, with_inverted=True, run_optimize=run_optimize, sorted_runs=True)  # runs
    v, idv, dv = H.random_dict_column(rng, "v", n, 100000, value_stride=7)
    seg = S.SegmentData("c5", n, [p, q, r, v])
    P = Q.Pred
    inv = lambda c, d: Q.leaf(P.dict_range(c, d, d + 1, inverted=True))
    filters = [
        Q.and_(inv(0, 1), inv(1, 5), inv(2, 7)),
        Q.and_(inv(0, 0), inv(1, 63)),
        inv(2, 39),
        Q.and_(inv(0, 1), Q.leaf(P.dict_range(3, 0, 50000))),                                   # bitmap AND scan (applyAnd)
        Q.or_(inv(1, 1), inv(1, 2), Q.leaf(P.dict_set(1, [7, 9, 30], 64, inverted=True))),      # IN over postings
        Q.leaf(P.dict_range(1, 3, 4, exclusive=True, inverted=True)),                           # NOT_EQ flips over [0, numDocs)
        Q.and_(Q.leaf(P.dict_range(2, 3, 20, inverted=True)), inv(0, 1)),                       # range served by postings
    ]
    aggs = [(Q.COUNT, -1), (Q.SUM, 3), (Q.MAX, 3)]
    with engine.open(seg) as gseg:
        for i, flt in enumerate(filters):
            spec = Q.QuerySpec(aggs, filter=flt)
            H.assert_results_equal(gseg.execute(spec), oracle.execute(seg, spec), check_stats=False)
            gw, gc = gseg.filter_bitmap(Q.QuerySpec([], filter=flt))
            ow, oc = oracle.filter_bitmap(seg, Q.QuerySpec([], filter=flt))
            assert gc == oc and (gw == ow).all(), i
    m = (idp == 1) & (idq == 5) & (idr == 7)
    with engine.open(seg) as gseg:
        got = gseg.execute(Q.QuerySpec(aggs, filter=filters[0]))
    assert got.aggregations[0].count == int(m.sum())
    assert got.aggregations[1].sum_i64 == int(dv[idv[m]].astype(np.int64).sum())


def test_block_val_set_readers(engine):
    """ForwardIndexReader.readDictIds / Dictionary.readIntValues / readDoubleValues for arbitrary docIds
    (FixedBitSVForwardIndexReaderV2Test.java:76-110: sequential, sparse and tail docIds)."""
    rng = np.random.default_rng(5)
    n = 99999
    v, ids, dv = H.random_dict_column(rng, "v", n, 30000)
    raw_vals = rng.integers(-2 ** 31, 2 ** 31 - 1, n, dtype=np.int64).astype(np.int32)
    seg = S.SegmentData("spi", n, [v, S.Column.raw("r", raw_vals)])
    with engine.open(seg) as gseg:
        for doc_ids in (np.arange(n), np.arange(17, 17 + 10000), np.sort(rng.choice(n, 5000, replace=False)),
                        np.array([n - 2, n - 1]), np.array([0]), np.array([], dtype=np.int64)):
            doc_ids = doc_ids.astype(np.int32)
            assert (gseg.read_dict_ids(0, doc_ids) == ids[doc_ids]).all()
            assert (gseg.read_dict_ids(0, doc_ids) == oracle.read_dict_ids(v.fwd, v.bits, n, doc_ids)).all() if len(doc_ids) else True
            assert (gseg.read_int_values(0, doc_ids) == oracle.read_int_values(seg, 0, doc_ids)).all()
            assert (gseg.read_double_values(0, doc_ids) == dv[ids[doc_ids]].astype(np.float64)).all()
            assert (gseg.read_int_values(1, doc_ids) == raw_vals[doc_ids]).all()
        with pytest.raises(_abi.PinotGpuError):
            gseg.read_int_values(0, np.array([n], dtype=np.int32))


def test_error_reporting(engine):
    rng = np.random.default_rng(1)
    v, _, _ = H.random_dict_column(rng, "v", 1000, 10)
    seg = S.SegmentData("err", 1000, [v])
    with engine.open(seg) as gseg:
        with pytest.raises(_abi.PinotGpuError) as e:
            gseg.execute(Q.QuerySpec([(Q.SUM, 3)]))
        assert e.value.status == _abi.PG_ERR_INVALID_ARGUMENT and "column" in str(e.value)
        with pytest.raises(_abi.PinotGpuError):
            gseg.execute(Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.raw_range(0, 0, 5))))
    bad = S.SegmentData("bad", 1001, [v])   # forward index size does not match numDocs
    with pytest.raises(_abi.PinotGpuError) as e:
        engine.open(bad)
    assert "expected" in str(e.value)


def test_concurrent_queries_on_one_segment(engine):
    """pg_execute is re-entrant per handle: combine worker threads run different queries on the same segment."""
    import threading
    rng = np.random.default_rng(8)
    n = 500000
    v, ids, dv = H.random_dict_column(rng, "v", n, 5000)
    seg = S.SegmentData("conc", n, [v])
    specs = [Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(0, lo, lo + 1000))) for lo in range(0, 4000, 500)]
    want = [oracle.execute(seg, s) for s in specs]
    errors = []
    with engine.open(seg) as gseg:
        def worker(i):
            try:
                for _ in range(5):
                    H.assert_results_equal(gseg.execute(specs[i]), want[i])
            except Exception as ex:  # noqa: BLE001
                errors.append(ex)
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(specs))]
        [t.start() for t in threads]
        [t.join() for t in threads]
    assert not errors, errors


@pytest.mark.parametrize("mode", ["1", "0"])
def test_value_plane_and_dictionary_gather_paths_agree(mode, monkeypatch):
    """SUM through the device-built value plane (PINOT_GPU_VALUE_PLANE=1) and through per-row dictionary gathers (=0)
    must both equal the oracle, for narrow, wide (31-bit range) and negative-valued dictionaries, same-column filters
    (the range is evaluated on the plane) and group-by."""
    import torch  # noqa: F401
    from pinot_amd.engine import Engine
    monkeypatch.setenv("PINOT_GPU_VALUE_PLANE", mode)
    eng = Engine(device_id=0, time_kernels=True)   # pg_init re-reads the environment
    try:
        rng = np.random.default_rng(12)
        n = 150001
        narrow, ids_n, dv_n = H.random_dict_column(rng, "narrow", n, 70000, value_stride=3)
        wide_values = np.sort(rng.choice(np.arange(-2 ** 31, 2 ** 31 - 1, 65537, dtype=np.int64), 5000, replace=False)).astype(np.int32)
        ids_w = rng.integers(0, 5000, n).astype(np.int32)
        wide = S.Column.from_dict_ids("wide", wide_values, ids_w)
        single = S.Column.from_dict_ids("single", np.array([-17], dtype=np.int32), np.zeros(n, dtype=np.int32))
        k, idk, _ = H.random_dict_column(rng, "k", n, 300)
        seg = S.SegmentData("planes", n, [narrow, wide, single, k])
        aggs = [(Q.SUM, 0), (Q.AVG, 1), (Q.SUM, 2), (Q.MIN, 0), (Q.MAX, 1), (Q.COUNT, -1), (Q.MIN, 1), (Q.MAX, 0)]
        filters = [None, Q.leaf(Q.Pred.dict_range(0, 20000, 50000)), Q.leaf(Q.Pred.dict_range(1, 100, 4000)),
                   Q.and_(Q.leaf(Q.Pred.dict_range(0, 1000, 69000)), Q.leaf(Q.Pred.dict_set(1, list(range(0, 5000, 3)), 5000))),
                   Q.leaf(Q.Pred.dict_range(3, 0, 3))]
        with eng.open(seg) as gseg:
            for flt in filters:
                spec = Q.QuerySpec(aggs, filter=flt)
                H.assert_results_equal(gseg.execute(spec), oracle.execute(seg, spec))
                gspec = Q.QuerySpec(aggs, filter=flt, group_by=[3])
                H.assert_results_equal(gseg.execute(gspec), oracle.execute(seg, gspec))
        got = None
        with eng.open(seg) as gseg:
            got = gseg.execute(Q.QuerySpec([(Q.SUM, 1)]))
        assert got.aggregations[0].sum_i64 == int(wide_values[ids_w].astype(np.int64).sum())
    finally:
        monkeypatch.delenv("PINOT_GPU_VALUE_PLANE")
        Engine(device_id=0, time_kernels=True)


def test_doc_range_leaves(engine):
    """Sorted-column predicates arrive as docId ranges (PG_PRED_DOC_RANGE): no column is read for them.  Exercised in the
    lane-private kernel (plane / dictId aggregations) and in the LDS-staged one (a raw aggregation column forces it)."""
    rng = np.random.default_rng(23)
    n = 70_001
    v = rng.integers(0, 50_000, n).astype(np.int32)
    f = rng.integers(0, 100, n).astype(np.int32)
    seg = S.SegmentData("docrange", n, [S.Column.dict_encoded("v", v), S.Column.dict_encoded("f", f), S.Column.raw("r", v)])
    flt = H.range_pred(seg, "f", upper=30, upper_inclusive=False)
    with engine.open(seg) as g:
        for lo, hi, excl in ((0, n - 1, False), (100, 100, False), (2047, 2049, False), (1024, 4095, False), (5000, 60_000, False),
                             (5000, 60_000, True), (0, 0, True), (n - 5, n + 100, False), (-7, 3, False), (10, 9, False), (2048, 2048 + 31, False)):
            dr = Q.leaf(Q.Pred.doc_range(lo, hi, exclusive=excl))
            for filt in (dr, Q.and_(Q.leaf(flt), dr), Q.or_(dr, Q.leaf(flt)), Q.not_(dr)):
                for aggs in ([(Q.COUNT, -1), (Q.SUM, 0), (Q.MAX, 0)], [(Q.COUNT, -1), (Q.SUM, 2), (Q.MIN, 2)]):
                    spec = Q.QuerySpec(aggs, filter=filt)
                    H.assert_results_equal(g.execute(spec), oracle.execute(seg, spec), check_stats=False)
            spec = Q.QuerySpec([(Q.COUNT, -1)], filter=dr)
            words, card = g.filter_bitmap(spec)
            owords, ocard = oracle.filter_bitmap(seg, spec)
            assert card == ocard and np.array_equal(words, owords)
        spec = Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0)], filter=Q.leaf(Q.Pred.doc_range(5000, 60_000)), group_by=[1])
        H.assert_results_equal(g.execute(spec), oracle.execute(seg, spec), check_stats=False)
