"""Shared builders for the null-handling tests: the reference's literal tables (tests/golden/null_handling_kats.json) as segments and
queries, a reducer for the two-instance cases, and a per-doc restatement of the filter rules used to cross-check random trees."""
import json
import math
import os

import numpy as np

from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

DTYPES = {"INT": np.int32, "LONG": np.int64, "FLOAT": np.float32, "DOUBLE": np.float64}
# FieldSpec.getDefaultNullValue (pinot-spi/src/main/java/org/apache/pinot/spi/data/FieldSpec.java:73-90)
DEFAULT_NULL = {
    ("DIMENSION", "INT"): -2 ** 31, ("DIMENSION", "LONG"): -2 ** 63, ("DIMENSION", "FLOAT"): -math.inf, ("DIMENSION", "DOUBLE"): -math.inf,
    ("METRIC", "INT"): 0, ("METRIC", "LONG"): 0, ("METRIC", "FLOAT"): 0.0, ("METRIC", "DOUBLE"): 0.0,
}
FUNCTIONS = {"COUNT": Q.COUNT, "SUM": Q.SUM, "MIN": Q.MIN, "MAX": Q.MAX, "AVG": Q.AVG}


def load_kats():
    with open(os.path.join(H.GOLDEN_DIR, "null_handling_kats.json")) as f:
        return json.load(f)


def nullable_column(name, values, data_type, field_type, raw=False, with_inverted=False):
    """What the segment creator stores for a nullable column: the default null value in the forward index / dictionary plus the null
    value vector (SegmentColumnarIndexCreator.indexRow -> NullValueVectorCreator.setNull)."""
    default = DEFAULT_NULL[(field_type, data_type)]
    mask = np.array([v is None for v in values], dtype=bool)
    stored = np.array([default if v is None else v for v in values], dtype=DTYPES[data_type])
    col = S.Column.raw_typed(name, stored) if raw else S.Column.dict_encoded_typed(name, stored, with_inverted=with_inverted)
    return col.with_nulls(mask)


def reduce_partials(function, partials, null_handling):
    """Broker reduce of the per-segment intermediate results (AggregationFunction.merge + extractFinalResult): with null handling a
    segment whose holder stayed null (count == 0) contributes nothing and the result is null when every segment did."""
    if null_handling:
        partials = [p for p in partials if p.count > 0]
        if not partials:
            return None
    if function == Q.SUM:
        return sum(p.sum for p in partials)
    if function == Q.MIN:
        return min(p.min for p in partials)
    if function == Q.MAX:
        return max(p.max for p in partials)
    if function == Q.AVG:
        return sum(p.sum for p in partials) / sum(p.count for p in partials)
    return sum(p.count for p in partials)


def column_has_nulls(seg, ci):
    return seg.columns[ci].null_vector is not None


def leaf_for(seg, spec, null_handling=True, inverted=False):
    """["LT"|"LE"|"GT"|"GE"|"EQ"|"NE"|"IN"|"NOT_IN"|"IS_NULL"|"IS_NOT_NULL", column, value(s)] lowered the way the reference's predicate
    evaluators lower it; an always-true predicate on a column with nulls becomes IS NOT NULL under null handling
    (FilterOperatorUtils.java:78-86)."""
    op, name = spec[0], spec[1]
    ci = seg.column_index(name)
    col = seg.columns[ci]
    if op in ("IS_NULL", "IS_NOT_NULL"):
        return Q.Pred.is_null(ci, exclusive=op == "IS_NOT_NULL")
    inverted = inverted and col.inverted is not None
    if col.encoding != 0:      # raw INT column
        v = spec[2]
        lo, hi = {"LT": (-2 ** 31, v - 1), "LE": (-2 ** 31, v), "GT": (v + 1, 2 ** 31 - 1), "GE": (v, 2 ** 31 - 1), "EQ": (v, v)}[op]
        return Q.Pred.raw_range(ci, lo, hi)
    if op == "EQ":
        p = H.eq_pred(seg, name, spec[2], inverted=inverted)
    elif op == "NE":
        p = H.eq_pred(seg, name, spec[2], exclusive=True, inverted=inverted)
    elif op in ("IN", "NOT_IN"):
        p = H.in_pred(seg, name, spec[2], exclusive=op == "NOT_IN", inverted=inverted)
    else:
        kw = {"LT": dict(upper=spec[2], upper_inclusive=False), "LE": dict(upper=spec[2]), "GT": dict(lower=spec[2], lower_inclusive=False),
              "GE": dict(lower=spec[2])}[op]
        p = H.range_pred(seg, name, **kw)
    if null_handling and p.kind == Q.Pred.match_all().kind and not p.exclusive and column_has_nulls(seg, ci):
        return Q.Pred.is_null(ci, exclusive=True)
    return p


def tree_for(seg, spec, null_handling=True, inverted=False):
    if spec[0] == "AND":
        return Q.and_(*[tree_for(seg, s, null_handling, inverted) for s in spec[1:]])
    if spec[0] == "OR":
        return Q.or_(*[tree_for(seg, s, null_handling, inverted) for s in spec[1:]])
    if spec[0] == "NOT":
        return Q.not_(tree_for(seg, spec[1], null_handling, inverted))
    return Q.leaf(leaf_for(seg, spec, null_handling, inverted))


def reference_trues(spec, columns, nulls, num_docs, raw_columns=()):
    """Per-doc restatement of getTrues / getNulls / getFalses (BaseFilterOperator.java:85-113, BaseColumnFilterOperator.java:45-64,
    And/Or/NotFilterOperator) over plain numpy columns; `columns` hold the stored values, `nulls` the boolean null masks."""
    def tnf(s):
        op = s[0]
        if op in ("AND", "OR"):
            parts = [tnf(c) for c in s[1:]]
            t = np.ones(num_docs, bool) if op == "AND" else np.zeros(num_docs, bool)
            u = t.copy()
            for (ct, cn, _) in parts:
                t = (t & ct) if op == "AND" else (t | ct)
                u = (u & (ct | cn)) if op == "AND" else (u | (ct | cn))
            return t, np.zeros(num_docs, bool), ~u
        if op == "NOT":
            ct, _, cf = tnf(s[1])
            return cf, np.zeros(num_docs, bool), ct
        name = s[1]
        if op == "IS_NULL":
            t = nulls[name].copy()
            return t, np.zeros(num_docs, bool), ~t
        if op == "IS_NOT_NULL":
            t = ~nulls[name]
            return t, np.zeros(num_docs, bool), ~t
        v = columns[name]
        m = {"LT": lambda: v < s[2], "LE": lambda: v <= s[2], "GT": lambda: v > s[2], "GE": lambda: v >= s[2], "EQ": lambda: v == s[2],
             "NE": lambda: v != s[2], "IN": lambda: np.isin(v, s[2]), "NOT_IN": lambda: ~np.isin(v, s[2])}[op]()
        # a predicate with no possible match / no possible miss is an Empty / MatchAll operator without a null set -- except the
        # always-true one on a column with nulls, which becomes the IS NOT NULL bitmap operator (FilterOperatorUtils.java:76-90)
        # (dictionary columns only: a raw-value evaluator is never always true / false for these predicates)
        dict_values = np.unique(v)
        inner = {"LT": lambda d: d < s[2], "LE": lambda d: d <= s[2], "GT": lambda d: d > s[2], "GE": lambda d: d >= s[2], "EQ": lambda d: d == s[2],
                 "NE": lambda d: d != s[2], "IN": lambda d: np.isin(d, s[2]), "NOT_IN": lambda d: ~np.isin(d, s[2])}[op](dict_values)
        if name in raw_columns:
            inner = np.array([True, False])
        if not inner.any():
            return np.zeros(num_docs, bool), np.zeros(num_docs, bool), np.ones(num_docs, bool)
        if inner.all():
            t = ~nulls[name] if nulls[name].any() else np.ones(num_docs, bool)
            return t, np.zeros(num_docs, bool), ~t
        n = nulls[name]
        t = m & ~n
        return t, n.copy(), ~(t | n)
    return tnf(spec)[0]
