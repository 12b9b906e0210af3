"""The oracle's NotDocIdIterator restatement against the reference's own iterator test, call by call.

dociditerators/NotDocIdIteratorTest.java:31-104 drives a NotDocIdIterator over RangelessBitmapDocIdIterators and over an OrDocIdIterator
of three of them with a script of advance() / next() calls and asserts every returned docId.  The same scripts through the oracle's
iterator objects (pinot_oracle.c it_next / it_advance, IT_NOT over IT_BITMAP / IT_OR) -- the objects whose numEntriesScannedInFilter the
transducer's "NOT children" (pg_filter_fsm.h) is held against.  The scan-leaf case has no reference test: its entries are derived by hand
from SVScanDocIdIterator.java:76-112 for the same script (the derivation is in the test), the returned docIds are the reference's."""
from oracle import oracle

EOF = -1                                   # the oracle's PO_EOF (the reference's Constants.EOF)
N = -1                                     # script: next()
DOCS1 = [1, 4, 6, 10, 15, 17, 18, 20]
DOCS2 = [0, 1, 5, 8, 15, 18]
DOCS3 = [1, 2, 6, 13, 16, 19]
DOCS4 = [0, 1, 2, 3, 4, 5]
MIXED = [1, N, N, 7, 13, N, 18, 21, 26]                                                      # NotDocIdIteratorTest.java:54-62
MIXED_WANT = [2, 3, 5, 7, 13, 14, 19, 21, EOF]
ALL_NEXT_WANT = [0, 2, 3, 5, 7, 8, 9, 11, 12, 13, 14, 16, 19, 21, 22, 23, 24, EOF]           # :64-82


def test_not_over_a_bitmap_iterator_call_by_call():
    got, entries = oracle.not_iterator_script(0, [DOCS1], 25, MIXED)
    assert got == MIXED_WANT and entries == 0
    got, _ = oracle.not_iterator_script(0, [DOCS1], 25, [N] * len(ALL_NEXT_WANT))
    assert got == ALL_NEXT_WANT
    # :98-104: a child that covers every doc; one that leaves three
    assert oracle.not_iterator_script(0, [DOCS4], 6, [N])[0] == [EOF]
    assert oracle.not_iterator_script(0, [DOCS4], 9, [N, N, N])[0] == [6, 7, 8]


def test_not_over_an_or_iterator_call_by_call():
    # :84-96: NOT over OR(bitmap1, bitmap2, bitmap3) -- "OR result: [0, 1, 2, 4, 5, 6, 8, 10, 13, 15, 16, 17, 18, 19, 20]"
    got, entries = oracle.not_iterator_script(1, [DOCS1, DOCS2, DOCS3], 25, [N] * 11)
    assert got == [3, 7, 9, 11, 12, 14, 21, 22, 23, 24, EOF] and entries == 0


def test_not_over_a_scan_leaf_counts_batches_and_advances():
    """The same scripts over an SVScanDocIdIterator whose matches are DOCS1: the docIds cannot differ; the entries follow
    SVScanDocIdIterator.java:76-112 --
      constructor      next(): one batch [0, 25): 25 entries, all eight matches in the batch           25
      advance(1)       target = the known doc: nothing; next() hands 1 on, the batch has 4            +0
      next(), next()   3; then 4 is handed on, the batch has 6                                        +0
      advance(7)       7 > 6: the leaf advances doc by doc 7, 8, 9, 10                                +4   (the batch is dropped)
      advance(13)      13 > 10: 13, 14, 15                                                            +3
      next()           14                                                                             +0
      advance(18)      18 > 15: 18 matches at once (+1); 18 is handed on: next() scans [19, 25)       +1 +6
      advance(21)      21 > 20: 21, 22, 23, 24, end                                                   +4
      advance(26)      past the end                                                                   +0          = 43"""
    got, entries = oracle.not_iterator_script(2, [DOCS1], 25, MIXED)
    assert got == MIXED_WANT and entries == 43
    # next() only: the constructor's batch holds every match; the pull behind the last match finds no docs left
    got, entries = oracle.not_iterator_script(2, [DOCS1], 25, [N] * len(ALL_NEXT_WANT))
    assert got == ALL_NEXT_WANT and entries == 25
    # 600 docs, matches at 10 and 590: constructor one batch [0, 256) (256); 10 handed on -> batches [256, 512) and [512, 600) (256 + 88)
    got, entries = oracle.not_iterator_script(2, [[10, 590]], 600, [10, 12])
    assert got == [11, 12] and entries == 600
    # the same leaf, asked beyond the known doc first: advance(300) drops the batch and walks 300 .. 590 (291), 590 is not asked about
    got, entries = oracle.not_iterator_script(2, [[10, 590]], 600, [300, 400])
    assert got == [300, 400] and entries == 256 + 291


def test_and_and_or_iterators_call_by_call():
    """AndDocIdIteratorTest.java:32-55 and OrDocIdIteratorTest.java:32-57: the leap-frogging AND and the OR the NOT's neighbours are."""
    a1 = [0, 1, 2, 3, 5, 7, 10, 12, 13, 15, 16, 18, 20]
    a2 = [1, 2, 4, 5, 6, 7, 9, 11, 12, 13, 15, 16, 17, 19, 20]
    a3 = [0, 2, 3, 4, 7, 8, 10, 11, 13, 15, 16, 19, 20]
    assert oracle.not_iterator_script(3, [a1, a2, a3], 21, [N, N, 10, 16, N, N])[0] == [2, 7, 13, 16, 20, EOF]
    assert oracle.not_iterator_script(4, [DOCS1, DOCS2, DOCS3], 21, [1, N, N, 7, 13, N, 18, N, 21])[0] == [1, 2, 4, 8, 13, 15, 18, 19, EOF]
