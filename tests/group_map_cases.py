"""Segments and expectations for group-by key spaces above the array-based threshold (the reference's IntMapBasedHolder range)."""
import numpy as np

from pinot_amd import segment as S
import helpers as H


def wide_group_segment(rng, num_docs, cards=(700, 900), skew=False):
    """Key columns k0, k1[, k2] (product of cardinalities > 10 000), an INT metric v, a DOUBLE metric d and a filter column f."""
    cols, ids = [], []
    for i, card in enumerate(cards):
        _, kid, _ = H.random_dict_column(rng, "k%d" % i, num_docs, card, value_stride=3)
        if skew:
            kid = np.minimum(kid, rng.integers(0, card, num_docs)).astype(np.int32)
        col = S.Column.from_dict_ids("k%d" % i, (np.arange(card, dtype=np.int64) * 3 - 17).astype(np.int32), kid)
        cols.append(col)
        ids.append(kid.astype(np.int64))
    v = rng.integers(-1000, 100000, num_docs).astype(np.int32)
    d = rng.normal(0.0, 1e4, num_docs)      # |values| well under helpers.FP_VALUE_SCALE: the atomics add in any order
    f = rng.integers(0, 1000, num_docs).astype(np.int32)
    cols += [S.Column.dict_encoded("v", v), S.Column.dict_encoded_typed("d", d), S.Column.dict_encoded("f", f)]
    raw = np.zeros(num_docs, dtype=np.int64)
    mult = 1
    for kid, card in zip(ids, cards):
        raw += kid * mult
        mult *= card
    return S.SegmentData("wide_groups", num_docs, cols), raw, v.astype(np.int64), d, f


def admitted_keys(raw, mask, limit):
    """Raw keys the reference's IntGroupIdMap admits: the first `limit` distinct keys in docId order among the matching docs
    (DictionaryBasedGroupKeyGenerator.java:1022-1047)."""
    keys = raw[mask]
    _, first = np.unique(keys, return_index=True)
    first.sort()
    return set(int(k) for k in keys[first[:limit]])
