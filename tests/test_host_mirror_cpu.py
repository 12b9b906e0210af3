"""CPU tests of the C++ host mirror: SQL subset parser and predicate lowering against the oracle's restatement."""
import numpy as np
import pytest

from oracle import oracle
from pinot_amd import host
from pinot_amd import segment as S


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


def test_range_evaluator_cases_of_the_reference_test():
    """RangeOfflineDictionaryPredicateEvaluatorTest.java:35-250: a sorted dictionary of DICT_LEN = 10 entries whose insertion index of a bound
    is the bound itself (here: the values 0..9), every combination of inclusive / exclusive bounds, the boundaries ("*" when the range
    starts at dictId 0 inclusive or ends at DICT_LEN - 1 inclusive) and the zero range; getMatchingDictIds is [start, end)."""
    import ctypes as C
    import json
    lib = host._lib()
    lib.ph_lower_range_predicate.restype = C.c_void_p
    lib.ph_lower_range_predicate.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]
    dict_len = 10
    d = oracle.dict_write(np.arange(dict_len, dtype=np.int32))

    def evaluator(lower, incl_lower, upper, incl_upper):
        lo = "*" if (lower == 0 and incl_lower) else str(lower)               # createPredicate :262-272
        hi = "*" if (upper == dict_len - 1 and incl_upper) else str(upper)
        st = C.c_int32()
        ptr = lib.ph_lower_range_predicate(d.ctypes.data, dict_len, lo.encode(), int(incl_lower), hi.encode(), int(incl_upper), C.byref(st))
        assert st.value == 0 and ptr
        try:
            return json.loads(C.string_at(ptr).decode())
        finally:
            lib.ph_free(ptr)

    def check(ev, first, last, always_false=False):
        assert ev["alwaysFalse"] == always_false and not ev["alwaysTrue"] and ev["isRange"]
        matching = list(range(ev["start"], ev["end"]))
        assert matching == list(range(first, last + 1)) and ev["numMatchingItems"] == len(matching)      # verifyDictId
        for dict_id in range(-1, dict_len + 1):                                                         # applySV
            assert (ev["start"] <= dict_id < ev["end"]) == (first <= dict_id <= last)

    check(evaluator(2, True, 5, True), 2, 5)          # testRanges: [2, 5]
    check(evaluator(2, False, 5, True), 3, 5)         # (2, 5]
    check(evaluator(2, True, 5, False), 2, 4)         # [2, 5)
    check(evaluator(2, False, 5, False), 3, 4)        # (2, 5)
    check(evaluator(0, True, 5, False), 0, 4)         # testBoundaries: [0, 5)
    check(evaluator(0, True, 5, True), 0, 5)          # [0, 5]
    check(evaluator(6, True, dict_len - 1, True), 6, dict_len - 1)       # [6, DICT_LEN - 1]
    check(evaluator(6, False, dict_len - 1, True), 7, dict_len - 1)      # (6, DICT_LEN - 1]
    check(evaluator(4, False, 5, False), 5, 4, always_false=True)        # testZeroRange: (4, 5)
    whole = evaluator(0, True, dict_len - 1, True)                       # both unbounded: every dictId
    assert whole["alwaysTrue"] and (whole["start"], whole["end"]) == (0, dict_len)


def _plan_segment():
    n = 1000
    i = np.arange(n, dtype=np.int32)
    nullable = S.Column.dict_encoded("n", (i * 3) % 50).with_nulls(i % 17 == 0)
    cols = [S.Column.dict_encoded("s", i // 10, with_inverted=True), S.Column.dict_encoded("inv", i % 8, with_inverted=True),
            S.Column.dict_encoded("scan", (i * 7) % 100), nullable]
    return host.HostSegment(S.SegmentData("planSegment", n, cols), load=False)


def test_filter_operator_folding_and_priorities_of_the_reference_test():
    """FilterOperatorUtilsTest.java:47-200 through SQL: getAnd/Or/NotFilterOperator fold Empty and MatchAll children
    (testGetAndFilterOperator, testGetOrFilterOperator), and the children of an AND run by priority whatever their order in the query
    (testPriority: sorted, NOT(sorted) < bitmap < AND < OR, NOT(OR) < scan < unknown -- the inverted-index operator of this fork is
    none of the classes the reorder knows).  The plan is read back with ph_explain_filter; no device."""
    seg = _plan_segment()
    ex = lambda where: host.explain_filter(seg, "SELECT COUNT(*) FROM planSegment WHERE " + where)
    try:
        empty, match_all, regular = "inv = 999", "inv != 999", "scan > 5"            # EmptyFilterOperator, MatchAllFilterOperator, a scan
        scan = "SCAN(scan dictIds 6..99)"
        assert ex(empty) == "EMPTY" and ex(match_all) == "MATCH_ALL" and ex(regular) == scan
        assert ex(empty + " AND " + match_all) == "EMPTY" and ex(empty + " AND " + regular) == "EMPTY" and ex(match_all + " AND " + regular) == scan
        assert ex(empty + " OR " + match_all) == "MATCH_ALL" and ex(empty + " OR " + regular) == scan and ex(match_all + " OR " + regular) == "MATCH_ALL"
        assert ex("NOT " + empty) == "MATCH_ALL" and ex("NOT " + match_all) == "EMPTY" and ex("NOT " + regular) == "NOT(" + scan + ")"
        # priority classes, highest first; every operator of a class before every operator of a later class, in either query order
        classes = [[("s = 7", "SORTED(s docs 70..79)"), ("NOT s = 7", "NOT(SORTED(s docs 70..79))")],
                   [("n IS NULL", "BITMAP(n IS NULL)")],
                   [("(scan > 5 AND n IS NOT NULL)", "AND(BITMAP(n IS NOT NULL), SCAN(scan dictIds 6..99))")],
                   [("(scan > 5 OR inv = 3)", "OR(SCAN(scan dictIds 6..99), INVERTED(inv dictIds 3..3))"),
                    ("NOT (scan > 5 OR inv = 3)", "NOT(OR(SCAN(scan dictIds 6..99), INVERTED(inv dictIds 3..3)))")],
                   [("scan < 50", "SCAN(scan dictIds 0..49)")],
                   [("inv = 2", "INVERTED(inv dictIds 2..2)")]]
        for a in range(len(classes)):
            for high_sql, high in classes[a]:
                for b in range(a + 1, len(classes)):
                    for low_sql, low in classes[b]:
                        want = "AND(" + high + ", " + low + ")"
                        assert ex(low_sql + " AND " + high_sql) == want and ex(high_sql + " AND " + low_sql) == want
        # the sort is stable: two scans keep the query's order
        assert ex("scan < 50 AND scan > 5") == "AND(SCAN(scan dictIds 0..49), SCAN(scan dictIds 6..99))"
        assert ex("scan > 5 AND scan < 50") == "AND(SCAN(scan dictIds 6..99), SCAN(scan dictIds 0..49))"
        # sorted columns: IN / NOT IN are docId ranges of one sorted-index operator, adjacent dictIds merged
        assert ex("s IN (1, 2, 5)") == "OR(SORTED(s docs 10..29), SORTED(s docs 50..59))" and ex("s NOT IN (0, 99)") == "SORTED(s docs 10..989)"
    finally:
        seg.destroy()


@pytest.mark.parametrize("data_type, type_min, type_max", [(0, -2 ** 31, 2 ** 31 - 1), (1, -2 ** 63, 2 ** 63 - 1)])
def test_raw_range_evaluator_cases_of_the_reference_test(data_type, type_min, type_max):
    """NoDictionaryRangePredicateEvaluatorTest.java:35-140 (testIntPredicateEvaluator / testLongPredicateEvaluator): every bound shape of a
    range on a raw INT / LONG column, applied to -20..19 and to the type's extremes."""
    import ctypes as C
    import json
    lib = host._lib()
    lib.ph_lower_raw_range_predicate.restype = C.c_void_p
    lib.ph_lower_raw_range_predicate.argtypes = [C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]

    def evaluator(lower, incl_lower, upper, incl_upper):
        st = C.c_int32()
        ptr = lib.ph_lower_raw_range_predicate(data_type, str(lower).encode(), int(incl_lower), str(upper).encode(), int(incl_upper), C.byref(st))
        assert st.value == 0 and ptr, (lib.ph_last_error() or b"").decode()
        try:
            ev = json.loads(C.string_at(ptr).decode())
        finally:
            lib.ph_free(ptr)
        return lambda v: (not ev["alwaysFalse"]) and ev["rawLower"] <= v <= ev["rawUpper"]

    for bounds, want in ((( -10, True, 10, True), lambda i: -10 <= i <= 10), ((-10, False, 10, True), lambda i: -10 < i <= 10),
                         ((-10, False, 10, False), lambda i: -10 < i < 10), (("*", False, 10, True), lambda i: i <= 10),
                         (("*", False, 10, False), lambda i: i < 10), ((10, True, "*", True), lambda i: i >= 10),
                         ((10, False, "*", False), lambda i: i > 10), (("*", False, "*", False), lambda i: True)):
        apply_sv = evaluator(*bounds)
        assert all(apply_sv(i) == want(i) for i in range(-20, 20)), bounds
    unbounded = evaluator("*", False, "*", False)
    assert unbounded(type_min) and unbounded(type_max)
    open_extremes = evaluator(type_min, False, type_max, False)          # (MIN_VALUE, MAX_VALUE): everything but the extremes
    assert all(open_extremes(i) for i in range(-20, 20)) and not open_extremes(type_min) and not open_extremes(type_max)


def test_sql_parser_never_crashes_on_mutated_queries():
    """The SQL subset is a test vehicle, but it reads untrusted text: every mutation of valid queries must come back as a parsed query or as
    one of the three error classes (1 QueryException, 2 UnsupportedOperationException, 3 other std::exception), never a crash."""
    seeds = ["SELECT COUNT(*), SUM(a) FROM t WHERE a > 1 AND (b IN (1, 2, -3) OR NOT c BETWEEN 5 AND 9) AND d <> 'x''y' GROUP BY k1, k2 ORDER BY COUNT(*) DESC, k1 NULLS FIRST LIMIT 7",
             "SET enableNullHandling = true; SET minServerGroupTrimSize = 3; SELECT k, AVG(a) AS v FROM t WHERE a IS NOT NULL GROUP BY k ORDER BY v, k DESC",
             "SELECT SUM(a) FILTER (WHERE b > 3 AND c = 'x'), MAX(a) FROM t WHERE d NOT BETWEEN 1 AND 5 OR e NOT IN (1, 2) GROUP BY e LIMIT 0",
             "select column17, count(*) from testTable group by column17 order by Min(column6) desc, column17 limit 15"]
    tokens = ["SELECT", "FROM", "WHERE", "GROUP", "BY", "ORDER", "LIMIT", "AND", "OR", "NOT", "IN", "BETWEEN", "IS", "NULL", "NULLS", "FIRST", "LAST", "ASC", "DESC", "AS",
              "FILTER", "SET", "(", ")", ",", ";", "*", "=", "<>", "<", ">=", "'", "''", "1", "-1", "1e400", "99999999999999999999", "a", "k", "COUNT", "SUM", "DISTINCTCOUNT", "\t", "\x00"[:0], "é"]
    rng = np.random.default_rng(2024)
    checked = 0
    for seed_sql in seeds:
        words = seed_sql.split(" ")
        for _ in range(400):
            w = list(words)
            for _ in range(int(rng.integers(1, 4))):
                op, at = int(rng.integers(0, 4)), int(rng.integers(0, len(w)))
                if op == 0 and len(w) > 1:
                    del w[at]
                elif op == 1:
                    w.insert(at, tokens[int(rng.integers(0, len(tokens)))])
                elif op == 2:
                    w[at] = tokens[int(rng.integers(0, len(tokens)))]
                else:
                    w[at] = w[at][: int(rng.integers(0, len(w[at]) + 1))]
            sql = " ".join(w)
            try:
                host.parse_sql(sql)
            except host.HostError as e:
                assert e.status in (1, 2, 3), (sql, e.status)
            checked += 1
    assert checked == 1600


def test_gpu_devices_config_and_least_loaded_placement():
    """pinot.server.query.executor.gpu.devices and where segments go (GpuPlanMaker / GpuSegmentCache, Java and C++ twins): one server
    process drives every device of the node; equal-sized segments land on device s mod N in open order, unequal ones keep
    the devices' resident bytes balanced, and a dropped segment's bytes make room again."""
    from pinot_amd import host
    assert host.placement("0-7", [])["devices"] == list(range(8))
    assert host.placement("0,2,4", [])["devices"] == [0, 2, 4]
    assert host.placement(" 0-3, 6 ,2", [])["devices"] == [0, 1, 2, 3, 6]
    assert host.placement("5", [10, 10])["placement"] == [5, 5]
    for bad in ("", "3-1", "a", "-1", "1-x"):
        with pytest.raises(host.HostError):
            host.placement(bad, [])
    gb = 1 << 30
    assert host.placement("0-7", [3 * gb] * 16)["placement"] == [s % 8 for s in range(16)]
    assert host.placement("0-3", [8 * gb, 1 * gb, 1 * gb, 1 * gb, 1 * gb, 1 * gb, 1 * gb])["placement"] == [0, 1, 2, 3, 1, 2, 3]
    # device 2 loses its 5 GB segment: the next segments go there first
    assert host.placement("0-3", [5 * gb] * 4 + [-((2 << 48) | (5 * gb)), 2 * gb, 2 * gb, 2 * gb, 2 * gb])["placement"] == [0, 1, 2, 3, -1, 2, 2, 2, 0]
