"""CPU test: the oracle's restatement of the Long / ArrayMap group-key holders (one 128-bit mixed-radix key, first-appearance ids up to
numGroupsLimit) against a per-doc numpy / dict restatement of the same rules (DictionaryBasedGroupKeyGenerator.java:628-700, 808+)."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import query as Q
import hash_holder_cases as HC


@pytest.mark.parametrize("case", HC.cases(), ids=[c[0] for c in HC.cases()])
def test_oracle_long_and_array_map_holders(case):
    seg, ids, specs = HC.build(case)
    nk = len(ids)
    for spec in specs:
        got = oracle.execute(seg, spec)
        assert got.group_key_kind == case[3]
        mask = None
        if spec.filter is not None:
            words, _ = oracle.filter_bitmap(seg, Q.QuerySpec([], filter=spec.filter))
            mask = np.unpackbits(words.view(np.uint8), bitorder="little")[: seg.num_docs].astype(bool)
        want, scanned = HC.numpy_groups(seg, ids, spec, mask)
        assert got.stats[0] == scanned
        assert sorted(got.groups) == sorted(want)
        limit = spec.num_groups_limit if spec.num_groups_limit > 0 else 100000
        assert got.num_groups_limit_reached == (len(want) >= limit)
        assert got.group_id_upper_bound == limit
        # rows come in ascending raw-key order: the last column is the most significant digit
        assert got.group_keys == sorted(got.group_keys, key=lambda t: tuple(reversed(t)))
        if case[3] == 1:
            cards = [seg.columns[j].cardinality for j in range(nk)]
            for tup, raw in zip(got.group_keys, got.group_ids64):
                acc, mult = 0, 1
                for d, c in zip(tup, cards):
                    acc += d * mult; mult *= c
                assert raw == acc
        for key, docs in want.items():
            docs = np.asarray(docs)
            for a, (fn, col) in enumerate(spec.aggregations):
                v = got.groups[key][a]
                if fn == Q.COUNT:
                    assert v.count == len(docs)
                    continue
                vals = oracle.read_int_values(seg, col, docs.astype(np.int32)).astype(np.int64)
                if fn in (Q.SUM, Q.AVG):
                    assert v.sum_i64 == int(vals.sum()) and v.sum == float(vals.sum())
                if fn == Q.AVG:
                    assert v.count == len(docs)
                if fn == Q.MIN:
                    assert v.min == float(vals.min())
                if fn == Q.MAX:
                    assert v.max == float(vals.max())
