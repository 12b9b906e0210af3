"""Lowered query description (Python view of `pg_query`).

The filter is expressed the way the reference's PredicateEvaluators leave it after dictionary lookup:
dictId ranges / dictId sets on dictionary-encoded columns, inclusive value ranges on raw columns.  Lowering
SQL-level predicates (values) to this form is done by the C++ host mirror (libpinot_host.so); tests may also
build it directly.
"""
import ctypes as C

import numpy as np

from . import _abi


class Pred:
    def __init__(self, kind, column=0, lo=0, hi=0, dict_ids=None, cardinality=0, exclusive=False, inverted=False):
        self.kind = kind
        self.column = column
        self.lo = int(lo)
        self.hi = int(hi)
        self.exclusive = bool(exclusive)
        self.inverted = bool(inverted)
        self.set_words = None
        if kind == _abi.PG_PRED_DICT_SET:
            words = np.zeros((max(cardinality, 1) + 31) // 32, dtype=np.uint32)
            for d in dict_ids or []:
                words[d >> 5] |= np.uint32(1 << (d & 31))
            self.set_words = words

    @staticmethod
    def match_all():
        return Pred(_abi.PG_PRED_MATCH_ALL)

    @staticmethod
    def match_none():
        return Pred(_abi.PG_PRED_MATCH_NONE)

    @staticmethod
    def dict_range(column, start, end, exclusive=False, inverted=False):
        """startDictId <= dictId < endDictId (SortedDictionaryBasedRangePredicateEvaluator); EQ is [d, d + 1)."""
        return Pred(_abi.PG_PRED_DICT_RANGE, column, start, end, exclusive=exclusive, inverted=inverted)

    @staticmethod
    def dict_set(column, dict_ids, cardinality, exclusive=False, inverted=False):
        return Pred(_abi.PG_PRED_DICT_SET, column, dict_ids=list(dict_ids), cardinality=cardinality, exclusive=exclusive, inverted=inverted)

    @staticmethod
    def raw_range(column, lo, hi, exclusive=False):
        """lo <= value <= hi, both inclusive (IntRawValueBasedRangePredicateEvaluator)."""
        return Pred(_abi.PG_PRED_RAW_RANGE, column, lo, hi, exclusive=exclusive)

    @staticmethod
    def is_null(column, exclusive=False):
        """column IS NULL (exclusive=True: IS NOT NULL): BitmapBasedFilterOperator over the column's null value vector."""
        return Pred(_abi.PG_PRED_IS_NULL, column, 0, 0, exclusive=exclusive)

    @staticmethod
    def doc_range(first_doc, last_doc, exclusive=False):
        """first_doc <= docId <= last_doc: what SortedIndexBasedFilterOperator derives from a sorted column's [start, end] pairs."""
        return Pred(_abi.PG_PRED_DOC_RANGE, 0, first_doc, last_doc, exclusive=exclusive)

    @staticmethod
    def raw_range_f64(column, lo, hi, exclusive=False):
        """lo <= value <= hi on a raw FLOAT / DOUBLE column (Float / DoubleRawValueBasedRangePredicateEvaluator)."""
        return Pred(_abi.PG_PRED_RAW_RANGE, column, f64_bits(lo), f64_bits(hi), exclusive=exclusive)


def f64_bits(x):
    """IEEE-754 bit pattern of a double as a signed 64-bit integer (how RAW_RANGE bounds of FLOAT / DOUBLE columns travel)."""
    import struct
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


class Node:
    def __init__(self, op, children=(), pred=None):
        self.op = op
        self.children = list(children)
        self.pred = pred


def leaf(pred):
    return Node(_abi.PG_FILTER_LEAF, pred=pred)


def and_(*children):
    return Node(_abi.PG_FILTER_AND, children)


def or_(*children):
    return Node(_abi.PG_FILTER_OR, children)


def not_(child):
    return Node(_abi.PG_FILTER_NOT, [child])


COUNT, SUM, MIN, MAX, AVG = _abi.PG_AGG_COUNT, _abi.PG_AGG_SUM, _abi.PG_AGG_MIN, _abi.PG_AGG_MAX, _abi.PG_AGG_AVG


class QuerySpec:
    def __init__(self, aggregations, filter=None, group_by=(), null_handling=False, num_groups_limit=0, stats_upper_bound_ok=False):
        """aggregations: list of (function, column_index) with column_index -1 for COUNT(*).
        null_handling: the query option enableNullHandling=true (PG_QUERY_NULL_HANDLING).
        stats_upper_bound_ok: PG_QUERY_STATS_UPPER_BOUND_OK -- numEntriesScannedInFilter of a leap-frogging filter may be the upper bound."""
        self.null_handling = bool(null_handling)
        self.stats_upper_bound_ok = bool(stats_upper_bound_ok)
        self.num_groups_limit = int(num_groups_limit)
        self.aggregations = [(int(f), int(c)) for f, c in aggregations]
        self.filter = filter
        self.group_by = [int(c) for c in group_by]
        self._build()

    def _build(self):
        nodes, preds = [], []

        def walk(n):
            if n.op == _abi.PG_FILTER_LEAF:
                # one Pred OBJECT behind several leaves is ONE predicate of the query (pg_filter_node.predicate may repeat: the
                # transducer of numEntriesScannedInFilter takes its inputs per predicate, not per leaf)
                at = next((i for i, p in enumerate(preds) if p is n.pred), -1)
                if at < 0:
                    preds.append(n.pred)
                    at = len(preds) - 1
                nodes.append((n.op, at, 0))
            else:
                for ch in n.children:
                    walk(ch)
                nodes.append((n.op, -1, len(n.children)))

        if self.filter is not None:
            walk(self.filter)
        self._nodes = (_abi.pg_filter_node * max(len(nodes), 1))()
        for i, (op, p, k) in enumerate(nodes):
            self._nodes[i].op, self._nodes[i].predicate, self._nodes[i].num_children = op, p, k
        self.predicates = preds                  # Pred objects in predicate-index order
        self._preds = (_abi.pg_predicate * max(len(preds), 1))()
        self._keep = []
        for i, p in enumerate(preds):
            cp = self._preds[i]
            cp.kind, cp.column = p.kind, p.column
            cp.eval = _abi.PG_EVAL_INVERTED if p.inverted else _abi.PG_EVAL_SCAN
            cp.exclusive = 1 if p.exclusive else 0
            cp.lo, cp.hi = p.lo, p.hi
            if p.set_words is not None:
                self._keep.append(p.set_words)
                cp.set_words = p.set_words.ctypes.data_as(C.POINTER(C.c_uint32))
                cp.num_set_words = int(p.set_words.shape[0])
        self._aggs = (_abi.pg_aggregation * max(len(self.aggregations), 1))()
        for i, (f, c) in enumerate(self.aggregations):
            self._aggs[i].function, self._aggs[i].column = f, c
        self._groups = (C.c_int32 * max(len(self.group_by), 1))(*self.group_by)
        q = _abi.pg_query()
        q.filter = self._nodes
        q.num_filter_nodes = len(nodes)
        q.num_predicates = len(preds)
        q.predicates = self._preds
        q.aggregations = self._aggs
        q.num_aggregations = len(self.aggregations)
        q.num_group_by = len(self.group_by)
        q.group_by_columns = self._groups
        q.num_groups_limit = self.num_groups_limit
        q.flags = (_abi.PG_QUERY_NULL_HANDLING if self.null_handling else _abi.PG_QUERY_DEFAULT) | (_abi.PG_QUERY_STATS_UPPER_BOUND_OK if self.stats_upper_bound_ok else 0)
        self.c = q


class AggValue:
    __slots__ = ("count", "sum", "sum_i64", "sum_exact", "min", "max")

    def __init__(self, v):
        self.count, self.sum, self.sum_i64, self.sum_exact, self.min, self.max = (
            int(v.count), float(v.sum), int(v.sum_i64), bool(v.sum_exact), float(v.min), float(v.max))

    def intermediate(self, function):
        """The reference's intermediate result type: COUNT -> Long, SUM/MIN/MAX -> Double, AVG -> (sum, count)."""
        if function == COUNT:
            return self.count
        if function == SUM:
            return self.sum
        if function == MIN:
            return self.min
        if function == MAX:
            return self.max
        return (self.sum, self.count)

    def __repr__(self):
        return "AggValue(count=%d sum=%r sum_i64=%d min=%r max=%r)" % (self.count, self.sum, self.sum_i64, self.min, self.max)


class Result:
    """Python copy of a `pg_result` (the C result is freed by the caller right after conversion)."""

    def __init__(self, res, spec):
        self.functions = [f for f, _ in spec.aggregations]
        self.stats = (int(res.stats.num_docs_scanned), int(res.stats.num_entries_scanned_in_filter),
                      int(res.stats.num_entries_scanned_post_filter), int(res.stats.num_total_docs))
        self.device_ms = float(res.device_ms)
        self.dominant_kernel_ms = float(res.dominant_kernel_ms)
        self.dominant_kernel = _abi.KERNEL_NAMES.get(int(getattr(res, "dominant_kernel", -1)), "")
        self.filter_entries_exact = bool(res.filter_entries_exact)      # stats[1] is the reference's count, not an upper bound
        na = int(res.num_aggregations)
        self.aggregations = [AggValue(res.aggregations[a]) for a in range(na)] if res.aggregations else []
        self.groups = {}
        self.group_id_upper_bound = int(res.group_id_upper_bound)
        self.num_groups_limit_reached = bool(res.num_groups_limit_reached)
        # Keys: the int raw key (Array / IntMap holders), or -- when the raw key is beyond an int (Long / ArrayMap holders) -- the tuple
        # of the key's dictIds in group-by column order.  `group_keys` holds the dictId tuples of every kind, row by row.
        self.group_key_kind = int(getattr(res, "group_key_kind", 0))
        ng = len(spec.group_by)
        self.group_keys = []
        for g in range(int(res.num_groups)):
            tup = tuple(int(res.group_key_dict_ids[g * ng + j]) for j in range(ng)) if res.group_key_dict_ids else None
            self.group_keys.append(tup)
            gid = int(res.group_ids[g]) if self.group_key_kind == 0 else tup
            self.groups[gid] = [AggValue(res.group_aggregations[g * na + a]) for a in range(na)]
        self.group_ids64 = [int(res.group_ids64[g]) for g in range(int(res.num_groups))] if (self.group_key_kind == 1 and res.group_ids64) else None

    def intermediates(self):
        return [v.intermediate(f) for v, f in zip(self.aggregations, self.functions)]
