"""GPU parity of the container-wise inverted-index AND (index_and_kernel in pinot_amd/csrc/pg_kernels.h): AndDocIdSet.iterator's
index-based branch (core/operator/docidsets/AndDocIdSet.java:127-165), BitmapCollection's inverted members
(core/operator/filter/BitmapCollection.java:58-128) and FastFilteredCountOperator, against the oracle, bit for bit -- results, the
materialised docId bitmap (pg_filter_bitmap) and the execution statistics."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

pytestmark = pytest.mark.gpu

P = Q.Pred
inv = lambda c, d: Q.leaf(P.dict_range(c, d, d + 1, inverted=True))


def check(gseg, seg, spec, kernel=None):
    got = gseg.execute(spec)
    H.assert_results_equal(got, oracle.execute(seg, spec), check_stats=True)
    if kernel:
        assert got.dominant_kernel == kernel, got.dominant_kernel
    if spec.filter is not None:
        gw, gc = gseg.filter_bitmap(Q.QuerySpec([], filter=spec.filter))
        ow, oc = oracle.filter_bitmap(seg, Q.QuerySpec([], filter=spec.filter))
        assert gc == oc and (gw == ow).all()
    return got


@pytest.mark.parametrize("run_optimize", [True, False])
@pytest.mark.parametrize("n", [70_001, 400_009])
def test_and_of_postings_in_every_container_kind(engine, n, run_optimize):
    rng = np.random.default_rng(n)
    p, idp, _ = H.random_dict_column(rng, "p", n, 2, with_inverted=True, run_optimize=run_optimize)         # bitset containers
    q, idq, _ = H.random_dict_column(rng, "q", n, 64, with_inverted=True, run_optimize=run_optimize)        # array containers
    r, idr, _ = H.random_dict_column(rng, "r", n, 40, with_inverted=True, run_optimize=run_optimize, sorted_runs=True)   # runs
    w, idw, _ = H.random_dict_column(rng, "w", n, 3000, with_inverted=True, run_optimize=run_optimize)      # sparse: many windows without a container
    v, idv, dv = H.random_dict_column(rng, "v", n, 100000, value_stride=7)
    f, idf, _ = H.random_dict_column(rng, "f", n, 1000)
    seg = S.SegmentData("ia", n, [p, q, r, w, v, f])
    aggs = [(Q.COUNT, -1), (Q.SUM, 4), (Q.MAX, 4), (Q.MIN, 4)]
    filters = [
        Q.and_(inv(0, 1), inv(1, 5), inv(2, 7)),
        Q.and_(inv(0, 0), inv(1, 63)),
        Q.and_(inv(3, 17), inv(0, 1)),                                                                   # a posting with a handful of docs
        Q.and_(inv(3, 17), inv(3, 18)),                                                                  # disjoint postings: empty
        inv(2, 39),                                                                                      # one leaf alone
        inv(3, 2999),
        Q.and_(inv(0, 1), Q.leaf(P.dict_range(5, 0, 500))),                                              # index AND, then a scan leaf
        Q.and_(Q.leaf(P.dict_range(5, 0, 100)), inv(1, 9), inv(2, 30)),                                   # scan leaf written first in the query
        Q.and_(Q.leaf(P.dict_set(1, [7, 9, 30], 64, inverted=True)), inv(0, 1)),                         # IN: three postings OR-ed into one child
        Q.and_(Q.leaf(P.dict_range(2, 3, 20, inverted=True)), inv(0, 1)),                                # 17 postings: over the inline limit -> dense child
        Q.and_(Q.leaf(P.dict_range(1, 3, 4, exclusive=True, inverted=True)), inv(0, 0)),                  # NOT_EQ member: complement over [0, numDocs)
        Q.and_(Q.leaf(P.dict_set(1, [1, 2, 3], 64, exclusive=True, inverted=True)), Q.leaf(P.dict_range(2, 0, 40, exclusive=True, inverted=True)), inv(0, 1)),   # NOT IN, and NOT(everything)
        Q.leaf(P.dict_range(1, 3, 4, exclusive=True, inverted=True)),
        Q.and_(inv(0, 1), Q.or_(inv(1, 1), inv(1, 2))),                                                  # an OR child keeps the general path
    ]
    with engine.open(seg) as g:
        for flt in filters:
            check(g, seg, Q.QuerySpec(aggs, filter=flt))
            check(g, seg, Q.QuerySpec([(Q.SUM, 4), (Q.COUNT, -1)], filter=flt, group_by=[1]))
    m = (idp == 1) & (idq == 5) & (idr == 7)
    with engine.open(seg) as g:
        got = g.execute(Q.QuerySpec(aggs, filter=filters[0]))
    assert got.aggregations[0].count == int(m.sum())
    assert got.aggregations[1].sum_i64 == int(dv[idv[m]].astype(np.int64).sum())


def test_count_over_an_index_only_filter_scans_nothing(engine):
    """FastFilteredCountOperator: the cardinality of the and-ed bitmaps, statistics (count, 0, 0, totalDocs)."""
    rng = np.random.default_rng(3)
    n = 300_017
    p, idp, _ = H.random_dict_column(rng, "p", n, 16, with_inverted=True)
    q, idq, _ = H.random_dict_column(rng, "q", n, 64, with_inverted=True)
    seg = S.SegmentData("ffc", n, [p, q])
    with engine.open(seg) as g:
        got = check(g, seg, Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(inv(0, 3), inv(1, 5))), kernel="index_and_kernel")
        assert got.aggregations[0].count == int(((idp == 3) & (idq == 5)).sum())
        assert got.stats == (got.aggregations[0].count, 0, 0, n)
        got = check(g, seg, Q.QuerySpec([(Q.COUNT, -1), (Q.COUNT, -1)], filter=inv(1, 63)), kernel="index_and_kernel")
        assert got.aggregations[1].count == int((idq == 63).sum())
        check(g, seg, Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(inv(0, 3), Q.leaf(P.dict_range(0, 3, 4, exclusive=True, inverted=True)))), kernel=None)   # x AND NOT x


@pytest.mark.parametrize("n", [1, 63, 2048, 65_535, 65_536, 65_537, 131_072 + 5])
def test_window_edges(engine, n):
    rng = np.random.default_rng(n)
    card = min(5, n)
    p, idp, _ = H.random_dict_column(rng, "p", n, card, with_inverted=True)
    q, idq, _ = H.random_dict_column(rng, "q", n, min(3, n), with_inverted=True)
    seg = S.SegmentData("edge", n, [p, q])
    with engine.open(seg) as g:
        for flt in (Q.and_(inv(0, 0), inv(1, 0)), Q.and_(Q.leaf(P.dict_range(0, 0, 1, exclusive=True, inverted=True)), inv(1, 0)), inv(0, card - 1)):
            check(g, seg, Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0), (Q.MAX, 1)], filter=flt))


@pytest.mark.parametrize("n", [70_001, 2_300_017])
def test_sparse_and_aggregated_inside_index_and_kernel(engine, n):
    """Round 5: when the postings' sizes say the AND leaves a handful of docs per 65 536-doc window, index_and_kernel reads the survivors'
    values itself (IndexAndParams.gather_*): ONE launch for `SUM(v) WHERE p = a AND q = b AND r = c` -- no bitmap, no scan_sparse_kernel
    behind it (AndDocIdSet.java:127-172 + the projection of the surviving docs).  Same answers and statistics as the oracle; the denser
    filters of the same query shape still take scan_sparse_kernel."""
    rng = np.random.default_rng(n + 5)
    p, idp, _ = H.random_dict_column(rng, "p", n, 16, with_inverted=True)
    q, idq, _ = H.random_dict_column(rng, "q", n, 64, with_inverted=True)
    r, idr, _ = H.random_dict_column(rng, "r", n, 256, with_inverted=True)
    v, idv, dv = H.random_dict_column(rng, "v", n, 100000, value_stride=7)          # arithmetic progression: the dictId stream is the plane
    f, idf, _ = H.random_dict_column(rng, "f", n, 1000, value_stride=3)
    seg = S.SegmentData("ig", n, [p, q, r, v, f])
    sparse = [Q.and_(inv(0, 3), inv(1, 5), inv(2, 7)), Q.and_(inv(1, 60), inv(2, 255)), Q.and_(inv(0, 15), inv(1, 0), inv(2, 100)),
              Q.and_(inv(2, 1), inv(2, 2))]                                                                        # (disjoint: nothing survives)
    with engine.open(seg) as g:
        for flt in sparse:
            for aggs in ([(Q.SUM, 3)], [(Q.COUNT, -1), (Q.SUM, 3), (Q.MIN, 3), (Q.MAX, 3), (Q.AVG, 3)], [(Q.SUM, 3), (Q.MAX, 4)], [(Q.MIN, 4), (Q.COUNT, -1)]):
                got = check(g, seg, Q.QuerySpec(aggs, filter=flt), kernel="index_and_kernel")
                again = g.execute(Q.QuerySpec(aggs, filter=flt))                   # the counters were zeroed behind the first answer
                H.assert_results_equal(again, oracle.execute(seg, Q.QuerySpec(aggs, filter=flt)))
            check(g, seg, Q.QuerySpec([(Q.COUNT, -1)], filter=flt), kernel="index_and_kernel")       # COUNT(*): the cardinality counters alone
            # three aggregated columns: more than the kernel gathers -> the two-kernel path
            check(g, seg, Q.QuerySpec([(Q.SUM, 3), (Q.MAX, 4), (Q.MIN, 0)], filter=flt))
        dense = Q.and_(inv(0, 3), inv(1, 5))                                       # 1 / 1024 of the docs: 64 per window -> scan_sparse_kernel
        got = check(g, seg, Q.QuerySpec([(Q.SUM, 3)], filter=dense))
        assert got.dominant_kernel in ("scan_sparse_kernel", "index_and_kernel")
