"""Shared by the CPU (oracle vs numpy) and GPU (kernels vs oracle) tests of group-by key spaces beyond an int: the reference's
LongMapBasedHolder (product of the cardinalities fits a long) and ArrayMapBasedHolder (it does not) --
core/query/aggregation/groupby/DictionaryBasedGroupKeyGenerator.java:150-184, 628-700, 808+."""
import numpy as np

from pinot_amd import query as Q
from pinot_amd import segment as S


def big_card_column(name, n, cardinality, distinct, seed, stride=3):
    """A dictionary column with `cardinality` entries of which only `distinct` dictIds occur (spread over the whole range): the
    key SPACE is huge, the number of groups is not."""
    rng = np.random.default_rng(seed)
    used = np.unique(rng.integers(0, cardinality, distinct).astype(np.int64))
    used[0], used[-1] = 0, cardinality - 1                       # both ends of the dictionary occur
    ids = used[rng.integers(0, len(used), n)].astype(np.int32)
    values = (np.arange(cardinality, dtype=np.int64) * stride - 17).astype(np.int32)
    return S.Column.from_dict_ids(name, values, ids), ids


def cases():
    """(name, num_docs, [(cardinality, distinct)], expected kind)"""
    return [
        ("long-2cols", 60_011, [(70_000, 300), (60_000, 200)], 1),                  # 4.2e9 > Integer.MAX_VALUE
        ("long-3cols", 100_003, [(3_000, 40), (3_000, 50), (3_000, 30)], 1),          # 2.7e10
        ("array-3cols", 80_021, [(2_200_000, 60), (2_100_000, 50), (2_300_000, 40)], 2),   # 1.06e19 > Long.MAX_VALUE
        ("long-dense", 150_001, [(66_000, 66_000), (40_000, 5)], 1),                 # many groups: most docs have a group of their own
        # eight columns of ~2^21 entries each (2^168 raw keys): three chained first tables in front of the aggregating one
        ("array-8cols", 40_009, [(2_200_000, 9), (2_100_000, 7), (2_300_000, 5), (2_150_000, 6), (2_250_000, 4), (2_120_000, 5), (2_310_000, 3), (2_170_000, 4)], 2),
    ]


def build(case, seed=0):
    name, n, dims, kind = case
    cols, ids = [], []
    for j, (card, distinct) in enumerate(dims):
        c, i = big_card_column("k%d" % j, n, card, distinct, seed + 10 * j + 1)
        cols.append(c); ids.append(i)
    v = S.Column.synthetic_uniform("v", n, (np.arange(5000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=seed + 101)      # affine: value plane = dictIds
    w = S.Column.synthetic_uniform("w", n, np.arange(300, dtype=np.int32) * 11 - 900, seed=seed + 102)
    f = S.Column.synthetic_uniform("f", n, np.arange(100, dtype=np.int32), seed=seed + 103)
    seg = S.SegmentData("hash_" + name, n, cols + [v, w, f])
    nk = len(dims)
    specs = [Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, nk), (Q.MAX, nk + 1), (Q.MIN, nk)], group_by=list(range(nk))),
             Q.QuerySpec([(Q.SUM, nk + 1), (Q.AVG, nk)], filter=Q.leaf(Q.Pred.dict_range(nk + 2, 0, 37)), group_by=list(range(nk))),
             Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, nk)], group_by=list(range(nk)), num_groups_limit=50),                        # the limit binds: first keys in docId order
             Q.QuerySpec([(Q.MAX, nk)], filter=Q.leaf(Q.Pred.dict_range(nk + 2, 50, 100)), group_by=list(range(nk)), num_groups_limit=7)]
    return seg, ids, specs


def numpy_groups(seg, ids, spec, filter_mask):
    """{dictId tuple: (count, {agg index: value})} with the reference's semantics: keys admitted in docId order up to numGroupsLimit,
    docs of later keys dropped."""
    n = seg.num_docs
    limit = spec.num_groups_limit if spec.num_groups_limit > 0 else 100000
    docs = np.flatnonzero(filter_mask) if filter_mask is not None else np.arange(n)
    keys = np.stack([ids[j][docs] for j in range(len(ids))], axis=1)
    admitted = {}
    out = {}
    for row, doc in zip(map(tuple, keys.tolist()), docs.tolist()):
        if row not in admitted:
            if len(admitted) >= limit:
                continue
            admitted[row] = len(admitted)
            out[row] = []
        out[row].append(doc)
    return out, len(docs)
