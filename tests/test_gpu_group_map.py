"""GPU parity for group-by key spaces above the array-based threshold: one direct-indexed HBM table slot per raw key, device-side
compaction, and the reference's numGroupsLimit rule (the first keys in docId order survive)."""
import os

import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
import helpers as H
import group_map_cases as GM

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cards,num_docs", [((700, 900), 300000), ((40, 50, 60), 70001), ((4000, 4000), 200000)])
def test_wide_key_spaces_against_the_oracle(engine, cards, num_docs):
    rng = np.random.default_rng(len(cards) * 1000 + num_docs)
    seg, raw, v, d, f = GM.wide_group_segment(rng, num_docs, cards)
    ci = seg.column_index
    keys = [ci("k%d" % i) for i in range(len(cards))]
    agg_lists = [[(Q.COUNT, -1), (Q.SUM, ci("v")), (Q.MAX, ci("v")), (Q.MIN, ci("v")), (Q.AVG, ci("v"))],
                 [(Q.SUM, ci("d")), (Q.MAX, ci("d")), (Q.COUNT, -1)],
                 [(Q.COUNT, -1)]]
    filters = [None, Q.leaf(H.range_pred(seg, "f", upper=100, upper_inclusive=False)),
               Q.or_(Q.leaf(H.range_pred(seg, "f", lower=990)), Q.not_(Q.leaf(H.range_pred(seg, "v", lower=0))))]
    with engine.open(seg) as gseg:
        for aggs in agg_lists:
            for flt in filters:
                spec = Q.QuerySpec(aggs, filter=flt, group_by=keys)
                got, want = gseg.execute(spec), oracle.execute(seg, spec)
                H.assert_results_equal(got, want)
                assert got.group_id_upper_bound == int(np.prod(cards)) and got.num_groups_limit_reached == want.num_groups_limit_reached
                assert list(got.groups) == sorted(got.groups)          # ascending raw keys


@pytest.mark.parametrize("limit", [1, 1000, 15000])
def test_num_groups_limit_keeps_the_first_keys_in_doc_order(engine, limit):
    rng = np.random.default_rng(limit)
    seg, raw, v, d, f = GM.wide_group_segment(rng, 150000, cards=(300, 400), skew=True)
    ci = seg.column_index
    aggs = [(Q.COUNT, -1), (Q.SUM, ci("v")), (Q.MIN, ci("v"))]
    with engine.open(seg) as gseg:
        for flt, mask in ((None, np.ones(len(raw), bool)), (Q.leaf(H.range_pred(seg, "f", upper=500, upper_inclusive=False)), f < 500)):
            spec = Q.QuerySpec(aggs, filter=flt, group_by=[ci("k0"), ci("k1")], num_groups_limit=limit)
            got, want = gseg.execute(spec), oracle.execute(seg, spec)
            H.assert_results_equal(got, want)
            assert set(got.groups) == GM.admitted_keys(raw, mask, limit) and len(got.groups) == limit
            assert got.num_groups_limit_reached and want.num_groups_limit_reached
            assert got.stats[0] == int(mask.sum())               # the docs of refused keys were still scanned
        # a limit that is exactly the number of groups: nothing is dropped, the flag is still raised (>=)
        present = len(np.unique(raw))
        spec = Q.QuerySpec(aggs, group_by=[ci("k0"), ci("k1")], num_groups_limit=present)
        got = gseg.execute(spec)
        H.assert_results_equal(got, oracle.execute(seg, spec))
        assert len(got.groups) == present and got.num_groups_limit_reached


def test_partitioned_path_on_a_segment_large_enough_to_take_it(engine):
    """>= 4 Mi docs: the docs are first partitioned by key range and every partition is aggregated in LDS (pg_group_partition.h)."""
    # the environment switches that route group-bys elsewhere (used to test the fallbacks) turn the kernel-name checks off
    default_routing = not any(os.environ.get(k) for k in ("PINOT_GPU_GROUP_PARTITION", "PINOT_GPU_GROUP_PRIVATE", "PINOT_GPU_SCAN_PRIVATE"))
    rng = np.random.default_rng(77)
    seg, raw, v, d, f = GM.wide_group_segment(rng, 5_000_000, cards=(700, 900), skew=True)
    ci = seg.column_index
    keys = [ci("k0"), ci("k1")]
    agg_lists = [[(Q.COUNT, -1)],
                 [(Q.SUM, ci("v")), (Q.COUNT, -1)],
                 [(Q.SUM, ci("v")), (Q.MAX, ci("f")), (Q.AVG, ci("v"))],
                 [(Q.MIN, ci("v")), (Q.MAX, ci("v")), (Q.SUM, ci("f"))]]          # three accumulators: 2048-slot partitions
    filters = [None, Q.leaf(H.range_pred(seg, "f", upper=100, upper_inclusive=False)),
               Q.or_(Q.leaf(H.range_pred(seg, "f", lower=990)), Q.not_(Q.leaf(H.range_pred(seg, "v", lower=0))))]
    with engine.open(seg) as gseg:
        for aggs in agg_lists:
            for flt in filters:
                spec = Q.QuerySpec(aggs, filter=flt, group_by=keys, num_groups_limit=1_000_000)
                got, want = gseg.execute(spec), oracle.execute(seg, spec)
                H.assert_results_equal(got, want)
                assert got.dominant_kernel == "group_partition_scatter_kernel" or not default_routing
        # the limit rule on top of the partitioned table
        spec = Q.QuerySpec([(Q.SUM, ci("v"))], group_by=keys, num_groups_limit=5000)
        got = gseg.execute(spec)
        H.assert_results_equal(got, oracle.execute(seg, spec))
        assert len(got.groups) == 5000 and got.num_groups_limit_reached
        # a DOUBLE sum is outside the 32-bit record format: the direct HBM-atomic path takes it
        spec = Q.QuerySpec([(Q.SUM, ci("d"))], group_by=keys, num_groups_limit=1_000_000)
        got = gseg.execute(spec)
        H.assert_results_equal(got, oracle.execute(seg, spec))
        assert got.dominant_kernel != "group_partition_scatter_kernel"


@pytest.mark.parametrize("cards,num_docs", [((3000, 2500), 4_300_000), ((60_000, 30_000), 4_500_000), ((6000, 5000, 40), 4_400_000)])
def test_two_level_partitioning_above_two_million_keys(engine, cards, num_docs):
    """Key spaces of 7.5 M and 1.8 G raw keys (the upper IntMapBasedHolder range, DictionaryBasedGroupKeyGenerator.java:164-181): more fine
    partitions than one scatter pass addresses -- the docs are scattered by coarse partition, every coarse partition's records once more by
    fine partition, then aggregated in LDS as before (group_repartition_*_kernel, pg_group_partition.h).  COUNT (packed records), one
    unsigned input (packed with the slot when it fits), two / three inputs (separate value columns), with and without a filter; against
    the oracle, and against the direct HBM-atomic path the same key spaces took before."""
    default_routing = not any(os.environ.get(k) for k in ("PINOT_GPU_GROUP_PARTITION", "PINOT_GPU_GROUP_PRIVATE", "PINOT_GPU_SCAN_PRIVATE", "PINOT_GPU_PARTITION_TWO_LEVEL"))
    import hash_holder_cases as HC
    from pinot_amd import segment as S
    rng = np.random.default_rng(cards[0])
    # huge key SPACE, moderate number of groups (the dictIds that occur are spread over the whole range: many partitions, sparsely filled)
    # (three key columns: the third one's multiplier, 30 M, is beyond the 24-bit multiply of the narrow key spaces)
    kcols = [HC.big_card_column("k%d" % i, num_docs, c, (120, 60, 7)[i], seed=11 + i)[0] for i, c in enumerate(cards)]
    vv = rng.integers(-1000, 100000, num_docs).astype(np.int32)
    ff = rng.integers(0, 1000, num_docs).astype(np.int32)
    seg = S.SegmentData("two_level", num_docs, kcols + [S.Column.dict_encoded("v", vv), S.Column.dict_encoded("f", ff)])
    ci = seg.column_index
    keys = [ci("k%d" % i) for i in range(len(cards))]
    agg_lists = [[(Q.COUNT, -1)],
                 [(Q.MAX, ci("f")), (Q.COUNT, -1)],
                 [(Q.SUM, ci("v")), (Q.COUNT, -1)],
                 [(Q.SUM, ci("v")), (Q.MAX, ci("f")), (Q.AVG, ci("v"))],
                 [(Q.MIN, ci("v")), (Q.MAX, ci("v")), (Q.SUM, ci("f"))]]
    filters = [None, Q.leaf(H.range_pred(seg, "f", upper=100, upper_inclusive=False))]
    with engine.open(seg) as gseg:
        for aggs in agg_lists:
            for flt in filters:
                spec = Q.QuerySpec(aggs, filter=flt, group_by=keys, num_groups_limit=10_000_000)
                got, want = gseg.execute(spec), oracle.execute(seg, spec)
                H.assert_results_equal(got, want)
                # (three accumulators halve the LDS slots of a fine partition: 1.8 G keys are then 2^11 fine partitions per coarse one,
                #  beyond kMaxFinePerCoarse -- that combination keeps the direct HBM atomics)
                expect_partition = not (len(aggs) == 3 and aggs[0][0] == Q.MIN and int(np.prod(cards)) > 2 ** 30)
                assert (got.dominant_kernel == "group_partition_scatter_kernel") == expect_partition or not default_routing
        spec = Q.QuerySpec([(Q.SUM, ci("v"))], group_by=keys, num_groups_limit=5000)
        got = gseg.execute(spec)
        H.assert_results_equal(got, oracle.execute(seg, spec))
        assert len(got.groups) == 5000 and got.num_groups_limit_reached
