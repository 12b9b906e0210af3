"""The oracle's filter evaluation against the known answers of the reference's {And,Or,Not}FilterOperatorTest (tests/filter_operator_kats.py)."""
import pytest

from oracle import oracle
from pinot_amd import query as Q
import filter_operator_kats as K


def docs(seg, tree, null_handling=False):
    words, card = oracle.filter_bitmap(seg, Q.QuerySpec([(Q.COUNT, -1)], filter=tree, null_handling=null_handling))
    return K.docs_of_bitmap(seg, words, card)


@pytest.mark.parametrize("check", [K.check_and_filter_operator_known_answers, K.check_or_filter_operator_known_answers,
                                   K.check_or_filter_operator_trues_and_falses_under_null_handling, K.check_not_filter_operator_known_answers,
                                   K.check_doc_id_iterator_sets, K.check_bitmap_collection_cardinalities])
def test_oracle_against_the_filter_operator_tests(check):
    check(docs)
