"""Value-plane residency (pg_engine.hip acquire_plane / PlaneRegistry): planes are built beside the queries (the first query never waits for
one: it runs the dictionary path and is right anyway), live under a process-wide HBM budget with least-recently-used eviction, and are
reported separately from the index buffers (pg_segment_plane_bytes)."""
import ctypes as C
import time

import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

pytestmark = pytest.mark.gpu


def irregular(rng, name, n, cardinality):
    values = np.sort(rng.choice(np.arange(-2_000_000, 2_000_000, dtype=np.int64), size=cardinality, replace=False)).astype(np.int32)
    ids = rng.integers(0, cardinality, n).astype(np.int32)
    ids[:cardinality] = np.arange(cardinality, dtype=np.int32)          # every dictionary entry is used
    return S.Column.from_dict_ids(name, values, ids)


def wait_for_plane(gseg, spec, more_than, tries=200):
    """Runs the query until the segment holds more plane bytes than `more_than` (the build runs on its own stream); every answer is returned."""
    answers = []
    for _ in range(tries):
        answers.append(gseg.execute(spec))
        if gseg.plane_bytes() > more_than:
            answers.append(gseg.execute(spec))                               # and once more: this one reads the plane
            break
        time.sleep(0.01)
    return answers


def test_planes_are_built_beside_the_queries_and_evicted_under_a_budget(engine):
    rng = np.random.default_rng(77)
    n = 600_011
    k = H.random_dict_column(rng, "k", n, 10)[0]
    cols = [k] + [irregular(rng, "x%d" % i, n, 20_000) for i in range(3)]
    seg = S.SegmentData("planes", n, cols)
    specs = [Q.QuerySpec([(Q.SUM, c), (Q.COUNT, -1)], group_by=[0]) for c in (1, 2, 3)]
    wants = [oracle.execute(seg, s) for s in specs]
    previous = C.c_uint64()
    lib = engine.lib
    with engine.open(seg) as g:
        assert g.plane_bytes() == 0
        base = g.device_bytes()
        # ---- first query: answered at once, the plane arrives afterwards
        first = g.execute(specs[0])
        H.assert_results_equal(first, wants[0])
        answers = wait_for_plane(g, specs[0], 0)
        for a in answers:
            H.assert_results_equal(a, wants[0])
        one_plane = g.plane_bytes()
        assert one_plane > 0 and g.device_bytes() == base + one_plane
        # ---- budget for one plane and a half: the three columns take turns, the least recently used plane goes
        assert lib.pg_set_plane_budget(int(1.5 * one_plane), C.byref(previous)) == 0
        try:
            for round_ in range(3):
                for c in (1, 2, 0):
                    held = g.plane_bytes()
                    for _ in range(6):                                              # before, while and after this column's plane is built
                        H.assert_results_equal(g.execute(specs[c]), wants[c])
                        time.sleep(0.02)
                    assert g.plane_bytes() <= int(1.5 * one_plane), (round_, c, held, g.plane_bytes())
            assert g.plane_bytes() == one_plane                                   # never two at once under this budget
            # ---- no budget at all: nothing is built, everything is still answered
            assert lib.pg_set_plane_budget(0, None) == 0
            for c in (0, 1, 2, 0):
                H.assert_results_equal(g.execute(specs[c]), wants[c])
        finally:
            assert lib.pg_set_plane_budget(previous.value, None) == 0


@pytest.mark.parametrize("kind", ["long", "double"])
def test_wide_plane_streams_an_eight_byte_dictionary_column(engine, kind):
    """SUM / AVG over a LONG / DOUBLE dictionary without a filter: the values are materialised once (8 bytes per doc, a raw column's image)
    and streamed by scan_private_typed_kernel instead of gathered per doc; MIN / MAX of the same column keep the dictId path."""
    rng = np.random.default_rng(5)
    n = 300_017
    if kind == "long":
        values = rng.integers(-2**40, 2**40, 5000).astype(np.int64)[rng.integers(0, 5000, n)]          # a range wider than 31 bits: 8-byte dictionary entries
    else:
        values = (rng.normal(size=4000) * 1e6)[rng.integers(0, 4000, n)].astype(np.float64)
    seg = S.SegmentData("wide", n, [S.Column.dict_encoded_typed("m", values), H.random_dict_column(rng, "f", n, 40)[0]])
    spec = Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1), (Q.AVG, 0)])
    want = oracle.execute(seg, spec)
    with engine.open(seg) as g:
        answers = wait_for_plane(g, spec, 0)
        for a in answers:
            H.assert_results_equal(a, want)
        assert g.plane_bytes() >= 8 * n and answers[-1].dominant_kernel == "scan_private_typed_kernel"
        if kind == "long":
            assert answers[-1].aggregations[0].sum_i64 == int(values.sum()) and answers[-1].aggregations[0].sum_exact
        # MIN / MAX on the column, or a filter: the plane is not used, the answers stay right
        for other in (Q.QuerySpec([(Q.SUM, 0), (Q.MAX, 0), (Q.MIN, 0)]), Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 13)))):
            H.assert_results_equal(g.execute(other), oracle.execute(seg, other))
