"""Shared by the CPU (oracle vs numpy) and GPU (kernels vs oracle) tests of GROUP BY over raw columns that have no int-range key image: FLOAT /
DOUBLE columns, INT / LONG columns spanning more than 31 bits.  NoDictionarySingleColumnGroupKeyGenerator keys all four stored types by
VALUE (core/query/aggregation/groupby/NoDictionarySingleColumnGroupKeyGenerator.java:100-135: Int / Long / Float / Double2IntOpenHashMap),
NoDictionaryMultiColumnGroupKeyGenerator by the tuple; ids by first appearance up to numGroupsLimit.  On the ABI such a column's digit is
the value's RANK among the column's distinct values (pg_group_key_info: is_offset = 2; pg_group_key_values)."""
import numpy as np

from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S


def cases():
    """(name, num_docs, key columns: ("double" | "float" | "wide-int" | "wide-long" | "int" | "dict", distinct values))"""
    return [
        ("single-double", 50_021, [("double", 300)]),
        ("single-float", 30_011, [("float", 120)]),
        ("wide-int", 40_009, [("wide-int", 200)]),
        ("wide-long", 20_003, [("wide-long", 90)]),
        ("double-and-dict", 60_013, [("double", 40), ("dict", 30)]),
        ("dict-int-and-double", 45_007, [("dict", 11), ("int", 64), ("double", 25)]),
        ("many-doubles", 70_001, [("double", 40_000)]),                        # above the LDS table: the partitioned / direct tables
        ("doubles-long-holder", 30_011, [("double", 3000), ("wide-long", 2500), ("dict", 700)]),      # 5.25e9 raw keys: LongMapBasedHolder
    ]


def special_doubles():
    return np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-310, -1e-310, 1.0, -1.0, np.finfo(np.float64).max, np.finfo(np.float64).min], dtype=np.float64)


def build(case, seed=0):
    """-> (SegmentData, per key column the per-doc key as int64 BITS of its identity (see key_identity), specs)"""
    name, n, keys = case
    rng = np.random.default_rng(seed + 17)
    cols, identities = [], []
    for j, (typ, distinct) in enumerate(keys):
        pick = rng.integers(0, distinct, n)
        if typ == "double":
            pool = np.concatenate([special_doubles(), rng.normal(0, 1e6, distinct)])[:distinct] if distinct >= 11 else rng.normal(0, 10, distinct)
            vals = pool[pick].astype(np.float64)
            cols.append(S.Column.raw_typed("k%d" % j, vals))
        elif typ == "float":
            with np.errstate(over="ignore"):           # (+-max of a double is +-inf as a float: wanted)
                pool = np.concatenate([special_doubles().astype(np.float32), rng.normal(0, 1e3, distinct).astype(np.float32)])[:distinct]
            vals = pool[pick].astype(np.float32)
            cols.append(S.Column.raw_typed("k%d" % j, vals))
        elif typ == "wide-int":
            pool = np.unique(np.concatenate([np.array([-(2 ** 31), 2 ** 31 - 1], dtype=np.int64), rng.integers(-(2 ** 31), 2 ** 31, distinct, dtype=np.int64)]))
            vals = pool[pick % len(pool)].astype(np.int32)
            cols.append(S.Column.raw("k%d" % j, vals))
        elif typ == "wide-long":
            pool = np.unique(np.concatenate([np.array([-(2 ** 63), 2 ** 63 - 1], dtype=np.int64), rng.integers(-(2 ** 62), 2 ** 62, distinct, dtype=np.int64)]))
            vals = pool[pick % len(pool)].astype(np.int64)
            cols.append(S.Column.raw_typed("k%d" % j, vals))
        elif typ == "int":
            vals = (pick.astype(np.int64) * 3 - 50).astype(np.int32)
            cols.append(S.Column.raw("k%d" % j, vals))
        else:
            values = (np.arange(distinct, dtype=np.int64) * 5 - 11).astype(np.int32)
            cols.append(S.Column.from_dict_ids("k%d" % j, values, pick.astype(np.int32)))
            vals = pick.astype(np.int64)
        identities.append(key_identity(vals))
    v = S.Column.synthetic_uniform("v", n, (np.arange(5000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=seed + 201)
    w = S.Column.synthetic_uniform("w", n, np.arange(300, dtype=np.int32) * 11 - 900, seed=seed + 202)
    f = S.Column.synthetic_uniform("f", n, np.arange(100, dtype=np.int32), seed=seed + 203)
    seg = S.SegmentData("rankkey_" + name, n, cols + [v, w, f])
    nk = len(keys)
    group_by = list(range(nk))
    specs = [Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, nk), (Q.MAX, nk + 1), (Q.MIN, nk)], group_by=group_by),
             Q.QuerySpec([(Q.SUM, nk + 1), (Q.AVG, nk)], filter=Q.leaf(Q.Pred.dict_range(nk + 2, 0, 37)), group_by=group_by),
             Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, nk)], group_by=group_by, num_groups_limit=23),        # the limit binds: first keys in docId order
             Q.QuerySpec([(Q.MAX, nk)], filter=Q.leaf(Q.Pred.dict_range(nk + 2, 50, 100)), group_by=group_by, num_groups_limit=7)]
    return seg, identities, specs


def key_identity(values):
    """What makes two key values the SAME key in the reference's maps, as int64: the long value; Double.doubleToLongBits of the (widened)
    double -- one NaN, -0.0 and 0.0 apart (fastutil's HashCommon.double2int / Double.doubleToLongBits equality)."""
    values = np.asarray(values)
    if np.issubdtype(values.dtype, np.floating):
        d = values.astype(np.float64)
        bits = d.view(np.int64).copy()
        bits[np.isnan(d)] = np.int64(0x7FF8000000000000)
        return bits
    return values.astype(np.int64)


def rank_values(seg, column):
    """The column's distinct values ascending in Double.compare's / Long.compare's order, as identities (what pg_group_key_values returns)."""
    c = seg.columns[column]
    from oracle import oracle
    docs = np.arange(seg.num_docs, dtype=np.int32)
    if c.stored_type in (_abi.PG_TYPE_FLOAT, _abi.PG_TYPE_DOUBLE):
        d = oracle.read_double_values(seg, column, docs)[0]
        ident = np.unique(key_identity(d))
        # order images: negative doubles descend in their bit patterns
        order = np.where(ident < 0, ~ident, ident | np.int64(-(2 ** 63))).view(np.uint64)
        return ident[np.argsort(order, kind="stable")]
    vals = oracle.read_double_values(seg, column, docs)[1] if c.stored_type == _abi.PG_TYPE_LONG else oracle.read_int_values(seg, column, docs).astype(np.int64)
    return np.unique(vals)


def is_rank_keyed(seg, column):
    c = seg.columns[column]
    if c.encoding == _abi.PG_FWD_FIXED_BIT_DICT:
        return False
    if c.stored_type in (_abi.PG_TYPE_FLOAT, _abi.PG_TYPE_DOUBLE):
        return True
    v = rank_values(seg, column)
    return int(v.max()) - int(v.min()) >= 0x7FFFFFFE


def key_tuples(result, seg, spec, values_of, base_of):
    """Result rows keyed by identity: rank-keyed columns through `values_of(column)` (identities, ascending), int-range raw columns through
    base_of(column) + digit, dictionary columns by dictId."""
    maps = []
    for c in spec.group_by:
        if is_rank_keyed(seg, c):
            vals = values_of(c)
            maps.append(lambda d, vals=vals: int(vals[d]))
        elif seg.columns[c].encoding != _abi.PG_FWD_FIXED_BIT_DICT:
            b = base_of(c)
            maps.append(lambda d, b=b: int(d) + b)
        else:
            maps.append(lambda d: int(d))
    out = {}
    for tup, vals in zip(result.group_keys, result.groups.values()):
        out[tuple(m(d) for m, d in zip(maps, tup))] = vals
    assert len(out) == len(result.group_keys)
    return out


def numpy_groups(identities, spec, filter_mask, num_docs):
    limit = spec.num_groups_limit if spec.num_groups_limit > 0 else 100000
    docs = np.flatnonzero(filter_mask) if filter_mask is not None else np.arange(num_docs)
    keys = np.stack([identities[j][docs] for j in range(len(identities))], axis=1)
    out = {}
    for row, doc in zip(map(tuple, keys.tolist()), docs.tolist()):
        if row not in out:
            if len(out) >= limit:
                continue
            out[row] = []
        out[row].append(doc)
    return out, len(docs)
