"""CPU test of the N>1 path: world_size-2 gloo processes, one segment per rank, host-side merge of the partials.
The per-rank "engine" here is the oracle (there is no GPU in this tier); what is under test is the sharding and merge
plumbing bench.py uses (pinot_amd/distributed.py): segment r on rank r, no data-path collective, merge on the host."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_queue):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from oracle import oracle
    from pinot_amd import distributed as D
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    n = 200003
    v = S.Column.synthetic_uniform("v", n, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=2 * rank + 1)
    f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2 * rank + 2)
    seg = S.SegmentData("c2b_%d" % rank, n, [v, f])
    res = oracle.execute(seg, Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))))
    per_rank = D.gather_partials([res.aggregations[0].sum_i64, res.aggregations[0].count], "cpu")
    merged = D.merge_sum_count(per_rank)
    elapsed = D.max_over_ranks(0.5 + rank, "cpu")
    dist.barrier()
    out_queue.put((rank, per_rank, merged, elapsed))
    dist.destroy_process_group()


def test_two_ranks_shard_segments_and_merge_on_host():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    results.sort()
    # every rank sees the same gathered partials and the same merge
    assert results[0][1] == results[1][1] and results[0][2] == results[1][2]
    assert results[0][3] == results[1][3] == 1.5          # MAX over ranks of the per-rank timings
    per_rank = results[0][1]
    assert per_rank[0] != per_rank[1]                     # different seeds -> different segments
    # single-process ground truth: the same two segments evaluated and merged serially
    from oracle import oracle
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    total_sum, total_count = 0, 0
    for r in range(world):
        n = 200003
        v = S.Column.synthetic_uniform("v", n, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=2 * r + 1)
        f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2 * r + 2)
        ids_f = S.synthetic_dict_ids(2 * r + 2, 0, n, 1000)
        ids_v = S.synthetic_dict_ids(2 * r + 1, 0, n, 100000)
        m = ids_f < 100
        assert per_rank[r] == [int((ids_v[m].astype(np.int64) * 7 + 3).sum()), int(m.sum())]
        total_sum += per_rank[r][0]
        total_count += per_rank[r][1]
    assert results[0][2] == (float(total_sum), total_count)


def _group_worker(rank, world, port, out_queue):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from oracle import oracle
    from pinot_amd import distributed as D
    from pinot_amd import query as Q
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    seg, _, _ = _group_segment(rank)
    res = oracle.execute(seg, Q.QuerySpec([(Q.SUM, 2), (Q.COUNT, -1), (Q.MIN, 2), (Q.MAX, 2), (Q.AVG, 2)], group_by=[0, 1]))
    rows = D.group_rows(res, seg, [0, 1])
    merged = D.merge_group_rows(D.gather_group_rows(rows))
    dist.barrier()
    out_queue.put((rank, sorted(res.groups), sorted((k, tuple(v)) for k, v in merged.items())))
    dist.destroy_process_group()


def _group_segment(rank):
    """Two key columns whose DICTIONARIES differ per rank: the same values sit under different dictIds (rank 1 has extra values in
    front), and some keys exist on one rank only."""
    from pinot_amd import segment as S
    rng = np.random.default_rng(100 + rank)
    n = 50_021
    kv = np.array([10, 20, 30, 40, 50], dtype=np.int32) if rank == 0 else np.array([-5, 0, 10, 30, 50, 70], dtype=np.int32)
    gv = np.array([1, 2, 3], dtype=np.int32) if rank == 0 else np.array([2, 3, 4, 5], dtype=np.int32)
    k = kv[rng.integers(0, kv.shape[0], n)]
    g = gv[rng.integers(0, gv.shape[0], n)]
    m = rng.integers(-1000, 1000, n).astype(np.int32)
    seg = S.SegmentData("g%d" % rank, n, [S.Column.dict_encoded("k", k), S.Column.dict_encoded("g", g), S.Column.dict_encoded("m", m)])
    return seg, (k, g), m


def test_group_by_partials_merge_on_key_values_not_dictids():
    """GroupByCombineOperator.java:132-147: the merge key is the tuple of VALUES.  The two ranks' dictionaries differ, so equal raw
    group ids mean different keys and equal keys have different raw ids: a merge keyed on dictIds / raw ids cannot pass this."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert results[0][2] == results[1][2]                  # every rank ends with the same merged table
    assert set(results[0][1]) & set(results[1][1])         # raw group ids DO collide across ranks ...
    # ground truth straight from the values
    truth = {}
    for r in range(world):
        _, (k, g), m = _group_segment(r)
        for key in set(zip(k.tolist(), g.tolist())):
            sel = (k == key[0]) & (g == key[1])
            c, s, mn, mx = int(sel.sum()), int(m[sel].astype(np.int64).sum()), float(m[sel].min()), float(m[sel].max())
            c0, s0, mn0, mx0 = truth.get(key, (0, 0, float("inf"), float("-inf")))
            truth[key] = (c0 + c, s0 + s, min(mn0, mn), max(mx0, mx))
    merged = dict(results[0][2])
    assert set(merged) == set(truth)
    only_rank1 = [key for key in truth if key[0] in (-5, 0, 70) or key[1] in (4, 5)]
    assert only_rank1                                       # ... and some keys exist on one rank only
    for key, (c, s, mn, mx) in truth.items():
        rows = merged[key]          # [(SUM), (COUNT), (MIN), (MAX), (AVG)] as (function, count, sum, sum_i64, min, max)
        assert rows[0][3] == s and rows[0][2] == float(s)
        assert rows[1][1] == c
        assert rows[2][4] == mn and rows[3][5] == mx
        assert rows[4][3] == s and rows[4][1] == c
