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
