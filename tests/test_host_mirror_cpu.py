"""CPU tests of the C++ host mirror: SQL subset parser and predicate lowering against the oracle's restatement."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import host


def test_parse_sql_shapes():
    q = host.parse_sql("SELECT COUNT(*), SUM(column1), MAX(column3), MIN(column6), AVG(column7) FROM testTable")
    assert q == {"table": "testTable", "aggregations": ["count(*)", "sum(column1)", "max(column3)", "min(column6)", "avg(column7)"],
                 "groupBy": [], "hasFilter": False}
    q = host.parse_sql("select sum(a) as s from t where a > 1 and (b in (1, 2, -3) or not c between 5 and 9) and d <> 'x''y' group by k1, k2")
    assert q["groupBy"] == ["k1", "k2"] and q["hasFilter"]
    q = host.parse_sql("SELECT SUM(a) FILTER (WHERE b > 3 AND c = 'x'), COUNT(*) FILTER(WHERE b > 3 AND c = 'x'), MAX(a) FROM t WHERE d < 5")
    assert len(q["aggregations"]) == 3 and q["aggregations"][2] == "max(a)" and q["hasFilter"]
    assert q["aggregations"][0].startswith("sum(a) FILTER(WHERE AND(") and q["aggregations"][1].startswith("count(*) FILTER(WHERE AND(")
    assert q["aggregations"][0].split("FILTER")[1] == q["aggregations"][1].split("FILTER")[1]       # same clause -> same swim lane
    q = host.parse_sql("SET enableNullHandling = true; SELECT COUNT(a), SUM(a) FROM t WHERE a IS NOT NULL AND (b IS NULL OR NOT c > 3)")
    assert q["nullHandling"] is True and q["hasFilter"] and q["aggregations"] == ["count(a)", "sum(a)"]
    assert "nullHandling" not in host.parse_sql("SET enableNullHandling = false; SELECT COUNT(*) FROM t")
    with pytest.raises(host.HostError):
        host.parse_sql("SELECT COUNT(*) FROM t WHERE a IS 3")
    for bad, status in (("SELECT a FROM t", 2), ("SET useStarTree = true; SELECT COUNT(*) FROM t", 2), ("SELECT SUM(a) FILTER (b > 3) FROM t", 1), ("SELECT SUM(a) FILTER (WHERE b > 3 FROM t", 1), ("SELECT SUM(a + 1) FROM t", 2), ("SELECT SUM(*) FROM t", 1), ("SELECT SUM(a) FROM", 1),
                        ("SELECT SUM(a) FROM t ORDER BY a", 2), ("SELECT SUM(a) FROM t WHERE a >", 1)):
        with pytest.raises(host.HostError) as e:
            host.parse_sql(bad)
        assert e.value.status == status, bad


def test_predicate_lowering_matches_the_oracle():
    values = np.array([-50, -3, 0, 7, 8, 100, 2 ** 31 - 1], dtype=np.int32)
    d = oracle.dict_write(values)
    n = len(values)
    cases = {
        "c BETWEEN 0 AND 8": dict(lower=0, upper=8),
        "c > 0": dict(lower=0, lower_inclusive=False),
        "c >= 1": dict(lower=1),
        "c < 8": dict(upper=8, upper_inclusive=False),
        "c <= 9": dict(upper=9),
        "c < -50": dict(upper=-50, upper_inclusive=False),
        "c > 100": dict(lower=100, lower_inclusive=False),
    }
    for sql, kw in cases.items():
        got = host.lower_predicate(sql, d, n)
        s, e = oracle.lower_range(d, n, **kw)
        assert got["isRange"] and (got["start"], got["end"]) == (s, e), sql
        assert got["alwaysFalse"] == (max(e - s, 0) == 0) and got["alwaysTrue"] == (e - s == n), sql
    eq = host.lower_predicate("c = 7", d, n)
    assert (eq["start"], eq["end"], eq["exclusive"]) == (3, 4, False)
    assert host.lower_predicate("c = 6", d, n)["alwaysFalse"]
    assert host.lower_predicate("c != 6", d, n)["alwaysTrue"]
    neq = host.lower_predicate("c <> 8", d, n)
    assert (neq["start"], neq["end"], neq["exclusive"]) == (4, 5, True)
    assert host.lower_predicate("c IN (8, 7, 7, 12345, -50)", d, n)["dictIds"] == [0, 3, 4]
    assert host.lower_predicate("c NOT IN (12345)", d, n)["alwaysTrue"]
    assert host.lower_predicate("c IN (-50, -3, 0, 7, 8, 100, 2147483647)", d, n)["alwaysTrue"]
    with pytest.raises(host.HostError) as e:
        host.lower_predicate("c = 'abc'", d, n)
    assert e.value.status == 1
    one = oracle.dict_write(np.array([5], dtype=np.int32))
    assert host.lower_predicate("c = 5", one, 1)["alwaysTrue"]     # EqualsPredicateEvaluatorFactory.java:103-105
    assert host.lower_predicate("c != 5", one, 1)["alwaysFalse"]
