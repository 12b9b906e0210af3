"""The kernels' filter evaluation (pg_filter_bitmap and COUNT(*)) against the known answers of the reference's {And,Or,Not}FilterOperatorTest
(tests/filter_operator_kats.py): bitmap leaves, scan leaves, inverted-index leaves, three-valued logic under null handling."""
import pytest

from pinot_amd import query as Q
import filter_operator_kats as K

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("check", [K.check_and_filter_operator_known_answers, K.check_or_filter_operator_known_answers,
                                   K.check_or_filter_operator_trues_and_falses_under_null_handling, K.check_not_filter_operator_known_answers,
                                   K.check_doc_id_iterator_sets, K.check_bitmap_collection_cardinalities])
def test_kernels_against_the_filter_operator_tests(engine, check):
    def docs(seg, tree, null_handling=False):
        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=tree, null_handling=null_handling)
        with engine.open(seg) as g:
            words, card = g.filter_bitmap(spec)
            assert g.execute(spec).aggregations[0].count == card
        return K.docs_of_bitmap(seg, words, card)
    check(docs)
