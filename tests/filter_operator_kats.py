"""Shared by test_oracle_filter_operator_kats.py (the oracle) and test_gpu_filter_operator_kats.py (the kernels): filter evaluation against the known answers of the reference's filter operator tests
(pinot-core/src/test/java/org/apache/pinot/core/operator/filter/{And,Or,Not}FilterOperatorTest.java): the docId lists of their
TestFilterOperators become columns whose value is 1 in those docs (and NULL in the tests' nullDocIds), the operators become the
filter tree, and getTrues() / getFalses() are the docs matching the tree / its negation under null handling."""
import numpy as np

from pinot_amd import query as Q
from pinot_amd import segment as S


def segment(num_docs, doc_id_lists, null_lists=None, inverted=()):
    cols = []
    for k, ids in enumerate(doc_id_lists):
        v = np.zeros(num_docs, dtype=np.int32)
        v[list(ids)] = 1
        col = S.Column.dict_encoded("f%d" % k, v, with_inverted=k in inverted)
        if null_lists and null_lists[k]:
            mask = np.zeros(num_docs, dtype=bool)
            mask[list(null_lists[k])] = True
            col = col.with_nulls(mask)
        cols.append(col)
    return S.SegmentData("ops", num_docs, cols)


def leaf(seg, k, inverted=False):
    card = seg.columns[k].cardinality          # {0, 1}: dictId 1 is the value 1; a column of all zeros has no dictId 1
    return Q.leaf(Q.Pred.dict_range(k, 1, 2, inverted=inverted)) if card == 2 else Q.leaf(Q.Pred.match_none())


def docs_of_bitmap(seg, words, card):
    out = [d for d in range(seg.num_docs) if (int(words[d >> 6]) >> (d & 63)) & 1]
    assert len(out) == card
    return out


def check_and_filter_operator_known_answers(docs):
    # testIntersectionForTwoLists / ThreeLists / testComplex (AndFilterOperatorTest.java:35-92)
    for inverted in ((), (0, 1, 2)):
        seg = segment(40, [[2, 3, 10, 15, 16, 28], [3, 6, 8, 20, 28]], inverted=inverted)
        assert docs(seg, Q.and_(leaf(seg, 0, 0 in inverted), leaf(seg, 1, 1 in inverted))) == [3, 28]
        seg = segment(40, [[2, 3, 6, 10, 15, 16, 28], [3, 6, 8, 20, 28], [1, 2, 3, 6, 30]], inverted=inverted)
        l = [leaf(seg, k, k in inverted) for k in range(3)]
        assert docs(seg, Q.and_(*l)) == [3, 6]
        assert docs(seg, Q.and_(Q.and_(l[0], l[1]), l[2])) == [3, 6]
    # testAndDocIdSetReordering (:94-135): multiples of 2, 3, 4, 5 among 10 000 docs, in either child order
    lists = [[j for j in range(10000) if j % i == 0] for i in range(2, 6)]
    seg = segment(10000, lists, inverted=(0, 1, 2, 3))
    for order in ((0, 1, 2, 3), (3, 2, 1, 0)):
        assert docs(seg, Q.and_(*[leaf(seg, k, True) for k in order]))[:4] == [0, 60, 120, 180]


def check_or_filter_operator_known_answers(docs):
    # testUnionForTwoLists / ThreeLists / testComplex (OrFilterOperatorTest.java:37-110): the sorted union
    a, b, c = [2, 3, 6, 10, 15, 16, 28], [3, 6, 8, 20, 28], [1, 2, 3, 6, 30]
    seg = segment(40, [[2, 3, 10, 15, 16, 28], b])
    assert docs(seg, Q.or_(leaf(seg, 0), leaf(seg, 1))) == sorted(set([2, 3, 10, 15, 16, 28]) | set(b))
    seg = segment(40, [a, b, c])
    l = [leaf(seg, k) for k in range(3)]
    assert docs(seg, Q.or_(*l)) == sorted(set(a) | set(b) | set(c)) == docs(seg, Q.or_(Q.or_(l[0], l[1]), l[2]))


def check_or_filter_operator_trues_and_falses_under_null_handling(docs):
    # testOrWithNull (:113-127): trues 0..3, falses 8, 9 (docs 4..7 are unknown)
    seg = segment(10, [[1, 2, 3], [0, 1, 2]], null_lists=[[4, 5, 6], [3, 4, 5, 6, 7]])
    tree = Q.or_(leaf(seg, 0), leaf(seg, 1))
    assert docs(seg, tree, True) == [0, 1, 2, 3] and docs(seg, Q.not_(tree), True) == [8, 9]
    # testOrWithNullHandlingButNoNullValues (:129-143)
    seg = segment(10, [[1, 2, 3], [0, 1, 2]])
    tree = Q.or_(leaf(seg, 0), leaf(seg, 1))
    assert docs(seg, tree, True) == [0, 1, 2, 3] and docs(seg, Q.not_(tree), True) == [4, 5, 6, 7, 8, 9]
    # testOrWithNullOneFilterIsEmpty (:145-157) / ...IsMatchAll (:159-171) / testOrWithNullTwoFiltersAreEmpty (:173-183)
    seg = segment(10, [[1, 2, 3]], null_lists=[[4, 5, 6]])
    tree = Q.or_(leaf(seg, 0), Q.leaf(Q.Pred.match_none()))
    assert docs(seg, tree, True) == [1, 2, 3] and docs(seg, Q.not_(tree), True) == [0, 7, 8, 9]
    tree = Q.or_(leaf(seg, 0), Q.leaf(Q.Pred.match_all()))
    assert docs(seg, tree, True) == list(range(10)) and docs(seg, Q.not_(tree), True) == []
    tree = Q.or_(Q.leaf(Q.Pred.match_none()), Q.leaf(Q.Pred.match_none()))
    assert docs(seg, tree, True) == [] and docs(seg, Q.not_(tree), True) == list(range(10))


def check_not_filter_operator_known_answers(docs):
    # testNotOperator (NotFilterOperatorTest.java:33-45): the complement within 30 docs
    ids = [2, 3, 10, 15, 16, 17, 18, 21, 22, 23, 24, 26, 28]
    seg = segment(30, [ids])
    assert docs(seg, Q.not_(leaf(seg, 0))) == [d for d in range(30) if d not in ids]
    # testNotWithNull (:47-58): trues 7..9, falses 0..3
    seg = segment(10, [[0, 1, 2, 3]], null_lists=[[4, 5, 6]])
    tree = Q.not_(leaf(seg, 0))
    assert docs(seg, tree, True) == [7, 8, 9] and docs(seg, Q.not_(tree), True) == [0, 1, 2, 3]
    # testNotEmptyFilterOperator (:60-68)
    seg = segment(5, [[0]])
    tree = Q.not_(Q.leaf(Q.Pred.match_none()))
    assert docs(seg, tree, True) == [0, 1, 2, 3, 4] and docs(seg, Q.not_(tree), True) == []


def check_doc_id_iterator_sets(docs):
    """operator/dociditerators/{And,Or,Not,Sorted}DocIdIteratorTest.java: the docId sets those iterators walk (their next / advance
    sequences visit exactly these docs)."""
    a = [[0, 1, 2, 3, 5, 7, 10, 12, 13, 15, 16, 18, 20], [1, 2, 4, 5, 6, 7, 9, 11, 12, 13, 15, 16, 17, 19, 20], [0, 2, 3, 4, 7, 8, 10, 11, 13, 15, 16, 19, 20]]
    seg = segment(21, a, inverted=(0, 1, 2))
    assert docs(seg, Q.and_(*[leaf(seg, k, True) for k in range(3)])) == [2, 7, 13, 15, 16, 20]          # AndDocIdIteratorTest :32-33
    assert docs(seg, Q.and_(*[leaf(seg, k) for k in range(3)])) == [2, 7, 13, 15, 16, 20]
    o = [[1, 4, 6, 10, 15, 17, 18, 20], [0, 1, 5, 8, 15, 18], [1, 2, 6, 13, 16, 19]]
    seg = segment(21, o, inverted=(0, 1, 2))
    want = [0, 1, 2, 4, 5, 6, 8, 10, 13, 15, 16, 17, 18, 19, 20]                                          # OrDocIdIteratorTest :32
    assert docs(seg, Q.or_(*[leaf(seg, k, True) for k in range(3)])) == want == docs(seg, Q.or_(*[leaf(seg, k) for k in range(3)]))
    seg = segment(25, [o[0]], inverted=(0,))                                                              # NotDocIdIteratorTest: 25 docs
    assert docs(seg, Q.not_(leaf(seg, 0, True))) == [0, 2, 3, 5, 7, 8, 9, 11, 12, 13, 14, 16, 19, 21, 22, 23, 24]
    seg = segment(40, [[0]])                                                                              # SortedDocIdIteratorTest: docId ranges
    R = lambda lo, hi: Q.leaf(Q.Pred.doc_range(lo, hi))
    assert docs(seg, R(1, 1)) == [1] and docs(seg, R(5, 15)) == list(range(5, 16))
    assert docs(seg, Q.or_(R(20, 25), R(30, 35))) == list(range(20, 26)) + list(range(30, 36))
    assert docs(seg, Q.or_(R(3, 3), R(8, 8), R(15, 15), R(20, 20))) == [3, 8, 15, 20]


# BitmapCollectionTest.java:30-205: (left docs, left inverted, right docs, right inverted, expected) over 10 docs
_AND_CARDINALITY = [((0, 5), False, (0, 4), False, 1), ((0, 5), False, (1, 4), False, 0), ((0, 5), False, (), False, 0), ((), False, (0, 5), False, 0),
                    ((), False, (), False, 0), ((0, 5), True, (0, 4), False, 1), ((0, 5), True, (1, 4), False, 2), ((0, 5), True, (), False, 0),
                    ((), True, (0, 5), False, 2), ((), True, (), False, 0), ((0, 5), False, (0, 4), True, 1), ((0, 5), False, (1, 4), True, 2),
                    ((0, 5), False, (), True, 2), ((), False, (), True, 0), ((), False, (0, 5), True, 0), ((0, 5), True, (0, 4), True, 7),
                    ((0, 5), True, (1, 4), True, 6), ((0, 5), True, (), True, 8), ((), True, (0, 5), True, 8), ((), True, (), True, 10)]
_OR_CARDINALITY = [((0, 5), False, (0, 4), False, 3), ((0, 5), False, (1, 4), False, 4), ((0, 5), False, (), False, 2), ((), False, (0, 5), False, 2),
                   ((), False, (), False, 0), ((0, 5), True, (0, 4), False, 9), ((0, 5), True, (1, 4), False, 8), ((0, 5), True, (), False, 8),
                   ((), True, (0, 5), False, 10), ((), True, (), False, 10), ((0, 5), False, (0, 4), True, 9), ((0, 5), False, (1, 4), True, 8),
                   ((0, 5), False, (), True, 10), ((), False, (0, 5), True, 8), ((), False, (), True, 10), ((0, 5), True, (0, 4), True, 9),
                   ((0, 5), True, (1, 4), True, 10), ((0, 5), True, (), True, 10), ((), True, (0, 5), True, 10), ((), True, (), True, 10)]


def check_bitmap_collection_cardinalities(docs):
    """andCardinality / orCardinality of two (possibly inverted) bitmaps: COUNT(*) of [NOT] left AND / OR [NOT] right over inverted-index leaves."""
    for cases, combine in ((_AND_CARDINALITY, Q.and_), (_OR_CARDINALITY, Q.or_)):
        for left, left_inverted, right, right_inverted, expected in cases:
            seg = segment(10, [left, right], inverted=(0, 1))
            l, r = leaf(seg, 0, True), leaf(seg, 1, True)
            tree = combine(Q.not_(l) if left_inverted else l, Q.not_(r) if right_inverted else r)
            assert len(docs(seg, tree)) == expected, (combine.__name__, left, left_inverted, right, right_inverted)
