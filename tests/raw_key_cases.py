"""Shared by the CPU (oracle vs numpy) and GPU (kernels vs oracle) tests of GROUP BY over raw (no-dictionary) INT / LONG columns: the
reference's NoDictionarySingleColumnGroupKeyGenerator / NoDictionaryMultiColumnGroupKeyGenerator
(core/query/aggregation/groupby/DefaultGroupByExecutor.java:106-121; NoDictionarySingleColumnGroupKeyGenerator.java:73-113, 240-247):
keys are VALUES, group ids are handed out in order of first appearance, new keys are refused once numGroupsLimit of them exist."""
import numpy as np

from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S


def cases():
    """(name, num_docs, key columns: list of ("int" | "long" | "dict", lowest value, value range, distinct values), expected key kind)"""
    return [
        ("single-int", 50_021, [("int", -5_000, 75_001, 300)], 0),
        ("small-range", 20_003, [("int", 17, 40, 40)], 0),                    # 40 possible keys: the ArrayBasedHolder range, yet numGroupsLimit binds
        ("int-and-dict", 60_013, [("int", -100, 9_001, 120), ("dict", 0, 30, 30)], 0),
        ("dict-and-int", 40_009, [("dict", 0, 11, 11), ("int", 1_000_000, 5_000, 64)], 0),
        ("single-long", 30_011, [("long", 10 ** 12, 200_000, 500)], 0),
        ("two-ints-long-holder", 45_007, [("int", -(2 ** 29), 2 ** 30, 200), ("int", 0, 70_000, 90)], 1),   # 2^30 x 70 000 > Integer.MAX_VALUE
    ]


def build(case, seed=0):
    """-> (SegmentData, per key column the true key VALUES (int64 arrays; dictionary columns: their dictIds), specs)"""
    name, n, keys, kind = case
    rng = np.random.default_rng(seed + 7)
    cols, key_values = [], []
    for j, (typ, lowest, span, distinct) in enumerate(keys):
        used = np.unique(rng.integers(0, span, distinct).astype(np.int64))
        used[0], used[-1] = 0, span - 1                             # both ends of the range occur: min and max are what the case says
        offs = used[rng.integers(0, len(used), n)]
        if typ == "dict":
            values = (np.arange(span, dtype=np.int64) * 5 - 11).astype(np.int32)
            cols.append(S.Column.from_dict_ids("k%d" % j, values, offs.astype(np.int32)))
            key_values.append(offs)
        elif typ == "int":
            cols.append(S.Column.raw("k%d" % j, (offs + lowest).astype(np.int32)))
            key_values.append(offs + lowest)
        else:
            cols.append(S.Column.raw_typed("k%d" % j, (offs + lowest).astype(np.int64)))
            key_values.append(offs + lowest)
    v = S.Column.synthetic_uniform("v", n, (np.arange(5000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=seed + 101)
    w = S.Column.synthetic_uniform("w", n, np.arange(300, dtype=np.int32) * 11 - 900, seed=seed + 102)
    f = S.Column.synthetic_uniform("f", n, np.arange(100, dtype=np.int32), seed=seed + 103)
    seg = S.SegmentData("rawkey_" + name, n, cols + [v, w, f])
    nk = len(keys)
    group_by = list(range(nk))
    specs = [Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, nk), (Q.MAX, nk + 1), (Q.MIN, nk)], group_by=group_by),
             Q.QuerySpec([(Q.SUM, nk + 1), (Q.AVG, nk)], filter=Q.leaf(Q.Pred.dict_range(nk + 2, 0, 37)), group_by=group_by),
             Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, nk)], group_by=group_by, num_groups_limit=23),        # the limit binds: first keys in docId order
             Q.QuerySpec([(Q.MAX, nk)], filter=Q.leaf(Q.Pred.dict_range(nk + 2, 50, 100)), group_by=group_by, num_groups_limit=7)]
    if keys[0][0] == "int":
        # the key column is also aggregated and filtered on (a raw range leaf): one caller column, two streams on the device
        specs.append(Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], filter=Q.leaf(Q.Pred.raw_range(0, lowest_of(keys[0]) + 1, lowest_of(keys[0]) + keys[0][2] // 2)), group_by=group_by))
    return seg, key_values, specs


def lowest_of(key):
    return key[1]


def is_raw(seg, column):
    return seg.columns[column].encoding != _abi.PG_FWD_FIXED_BIT_DICT


def key_tuples(result, seg, spec, base_of):
    """Result rows keyed by what the reference's generators key them by: the VALUE of a raw column (base + digit, pg_group_key_info), the
    dictId of a dictionary column.  -> {tuple: [AggValue...]} in row order."""
    bases = [base_of(c) if is_raw(seg, c) else 0 for c in spec.group_by]
    out = {}
    for tup, vals in zip(result.group_keys, result.groups.values()):
        out[tuple(int(d) + b for d, b in zip(tup, bases))] = vals
    assert len(out) == len(result.group_keys)
    return out


def numpy_groups(key_values, spec, filter_mask, num_docs):
    """{key tuple: [docIds]} with the reference's semantics: keys admitted in docId order up to numGroupsLimit, docs of later keys dropped."""
    limit = spec.num_groups_limit if spec.num_groups_limit > 0 else 100000
    docs = np.flatnonzero(filter_mask) if filter_mask is not None else np.arange(num_docs)
    keys = np.stack([key_values[j][docs] for j in range(len(key_values))], axis=1)
    out = {}
    for row, doc in zip(map(tuple, keys.tolist()), docs.tolist()):
        if row not in out:
            if len(out) >= limit:
                continue
            out[row] = []
        out[row].append(doc)
    return out, len(docs)
