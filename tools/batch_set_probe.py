"""pg_execute_batch over segments whose queries carry dictId-SET leaves (IN lists, NOT IN): the items' words ride in the batch's blob and the
items share the launch of their kind (round 6; rounds 3-5: a set leaf's upload tied the item to a context and a launch of its own).
Run by tests/test_gpu_batch.py with PINOT_GPU_BATCH_TRACE=1 (the library says on stderr how many items every shared launch carried); every
answer is held against the oracle and against pg_execute, twice (the second call through the plan cache).  One JSON line on stdout."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def main():
    import helpers as H
    from oracle import oracle
    from pinot_amd import _abi
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine

    engine = Engine(device_id=0)
    segs = []
    for s, n in enumerate([70_001, 1, 200_003, 65_536, 1_000_003, 131_073, 333_337, 2049, 500_000, 12_345, 777_777, 4096]):
        rng = np.random.default_rng(6100 + s)
        cols = [H.random_dict_column(rng, "a", n, 900)[0], H.random_dict_column(rng, "b", n, 200)[0], H.random_dict_column(rng, "c", n, 40)[0],
                S.Column.synthetic_uniform("v", n, (np.arange(3000, dtype=np.int64) * 7 + 3 + s).astype(np.int32), seed=5 * s + 1),
                S.Column.synthetic_uniform("w", n, np.sort(np.random.default_rng(s).choice(2 ** 30, 5000, replace=False)).astype(np.int32), seed=5 * s + 2)]
        segs.append(S.SegmentData("set%d" % s, n, cols))

    def in_list(col, card, seed, k, **kw):
        ids = np.sort(np.random.default_rng(seed).choice(card, k, replace=False))
        return Q.leaf(Q.Pred.dict_set(col, [int(x) for x in ids], card, **kw))

    shapes = {
        # the general lane-private body (lean_kind 0): two aggregated columns under an IN list
        # (a root AND of two scan leaves leap-frogs: its statistic needs a pass behind the kernel, and such an item keeps a launch of its own)
        "private": lambda s: Q.QuerySpec([(Q.SUM, 3), (Q.MAX, 0)], filter=in_list(0, 900, 10 + s, 300)),
        # two sets under an OR, one of them NOT IN
        "two-sets": lambda s: Q.QuerySpec([(Q.SUM, 3)], filter=Q.or_(in_list(0, 900, 20 + s, 100), in_list(1, 200, 30 + s, 170, exclusive=True))),
        # COUNT under a set over a narrow column (scan_narrow_batch_kernel)
        "narrow": lambda s: Q.QuerySpec([(Q.COUNT, -1)], filter=in_list(2, 40, 40 + s, 11)),
        # a group-by of the LDS-table form under an IN list (group_lds_batch_kernel)
        "group-by": lambda s: Q.QuerySpec([(Q.SUM, 3), (Q.MAX, 1)], filter=in_list(0, 900, 60 + s, 400), group_by=[2]),
        # a leap-frogging root AND whose caller takes the statistic's upper bound (PG_QUERY_STATS_UPPER_BOUND_OK): no pass behind the kernel, so it shares too
        "flagged-and": lambda s: Q.QuerySpec([(Q.SUM, 3), (Q.MAX, 0)], filter=Q.and_(in_list(0, 900, 70 + s, 300), Q.leaf(Q.Pred.dict_range(1, 0, 150))), stats_upper_bound_ok=True),
        # SUM through the LDS histogram (a dictionary without structure) under an IN list
        "hist": lambda s: Q.QuerySpec([(Q.SUM, 4)], filter=in_list(0, 900, 50 + s, 450)),
    }
    report = {"failed": [], "shapes": {}}
    opened = [engine.open(seg) for seg in segs]
    try:
        for name, make in shapes.items():
            specs = [make(s) for s in range(len(segs))]
            sys.stderr.write("== shape %s\n" % name)
            sys.stderr.flush()
            for rep in range(2):
                for s, (status, res) in enumerate(engine.execute_batch(opened, specs)):
                    ok = status == _abi.PG_OK
                    if ok:
                        try:
                            if name.startswith("flagged"):      # numEntriesScannedInFilter is the upper bound there, by request
                                plain = Q.QuerySpec(specs[s].aggregations, filter=specs[s].filter, group_by=specs[s].group_by)
                                exact = opened[s].execute(plain)
                                H.assert_results_equal(exact, oracle.execute(segs[s], plain))
                                assert not res.filter_entries_exact and res.stats[1] == 2 * segs[s].num_docs
                                assert (res.stats[0], res.stats[2], res.stats[3]) == (exact.stats[0], exact.stats[2], exact.stats[3])
                                assert [(a.count, a.sum_i64, a.min, a.max) for a in res.aggregations] == [(a.count, a.sum_i64, a.min, a.max) for a in exact.aggregations]
                            else:
                                H.assert_results_equal(res, oracle.execute(segs[s], specs[s]))
                            single = opened[s].execute(specs[s])
                            ok = res.stats == single.stats and [(a.count, a.sum_i64, a.min, a.max) for a in res.aggregations] == [(a.count, a.sum_i64, a.min, a.max) for a in single.aggregations]
                        except AssertionError:
                            ok = False
                    if not ok:
                        report["failed"].append([name, rep, s])
            report["shapes"][name] = len(segs)
    finally:
        [g.close() for g in opened]
    print(json.dumps(report))


if __name__ == "__main__":
    main()
