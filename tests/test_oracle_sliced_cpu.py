"""CPU test of bench.py's full-size checker: the oracle over equal row-range slices on several threads (oracle.execute_sliced) must
merge to exactly what one oracle pass over the whole segment gives -- aggregations, group-bys and docs scanned."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
from pinot_amd import segment as S


@pytest.mark.parametrize("n,threads", [(200_003, 3), (1_000_001, 8), (70_000, 16)])
def test_sliced_oracle_equals_one_pass(n, threads):
    rng = np.random.default_rng(n)
    dv = np.sort(rng.permutation(np.arange(-50_000, 50_000, dtype=np.int32) * 977)[:9000]).astype(np.int32)
    v = S.Column.from_dict_ids("v", dv, rng.integers(0, 9000, n).astype(np.int32))
    f = S.Column.from_dict_ids("f", np.arange(1000, dtype=np.int32), rng.integers(0, 1000, n).astype(np.int32))
    k = S.Column.from_dict_ids("k", np.arange(37, dtype=np.int32) * 5, rng.integers(0, 37, n).astype(np.int32))
    seg = S.SegmentData("sl", n, [v, f, k])
    flt = Q.leaf(Q.Pred.dict_range(1, 0, 300))
    for spec in (Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, 0), (Q.MIN, 0), (Q.MAX, 0), (Q.AVG, 0)], filter=flt),
                 Q.QuerySpec([(Q.SUM, 0), (Q.MAX, 1)], group_by=[2]),
                 Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=flt, group_by=[2])):
        whole = oracle.execute(seg, spec)
        merged = oracle.execute_sliced(seg, spec, threads=threads)
        assert merged["slices"] == max(1, min(threads, n // 65536))
        assert merged["docs_scanned"] == whole.stats[0]
        assert oracle.matches_sliced(whole, merged, [fn for fn, _ in spec.aggregations])
    # and the comparison is not vacuous
    other = oracle.execute(seg, Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 301))))
    assert not oracle.matches_sliced(other, oracle.execute_sliced(seg, Q.QuerySpec([(Q.SUM, 0)], filter=flt), threads=threads), [Q.SUM])
