"""CPU tests of the oracle's IntMapBasedHolder restatement (group-by key spaces above arrayBasedThreshold, numGroupsLimit)."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
import helpers as H
import group_map_cases as GM


@pytest.mark.parametrize("limit", [0, 500, 7])
def test_int_map_holder_against_numpy(limit):
    rng = np.random.default_rng(11)
    seg, raw, v, d, f = GM.wide_group_segment(rng, 40000, cards=(300, 400), skew=True)
    ci = seg.column_index
    aggs = [(Q.COUNT, -1), (Q.SUM, ci("v")), (Q.MAX, ci("v")), (Q.MIN, ci("v")), (Q.AVG, ci("v"))]
    for flt, mask in ((None, np.ones(len(raw), bool)), (Q.leaf(H.range_pred(seg, "f", upper=300, upper_inclusive=False)), f < 300)):
        res = oracle.execute(seg, Q.QuerySpec(aggs, filter=flt, group_by=[ci("k0"), ci("k1")], num_groups_limit=limit))
        present = np.unique(raw[mask])
        effective = limit if limit else 100000
        keep = GM.admitted_keys(raw, mask, effective)
        assert set(res.groups) == keep and len(keep) == min(len(present), effective)
        assert res.num_groups_limit_reached == (len(present) >= effective)
        assert res.stats[0] == int(mask.sum()) and res.stats[2] == 3 * int(mask.sum())
        for key in list(keep)[:200]:
            m = mask & (raw == key)
            c, s, mx, mn, avg = res.groups[key]
            assert c.count == m.sum() and s.sum_i64 == int(v[m].sum()) and mx.max == float(v[m].max()) and mn.min == float(v[m].min())
            assert avg.count == m.sum() and avg.sum_i64 == int(v[m].sum())
