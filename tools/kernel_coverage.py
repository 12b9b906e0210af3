#!/usr/bin/env python3
"""Kernel coverage table: one query (or call) per kernel instantiation the planner can pick, in the multi-tile regime.

Why: a kernel that was wrong from a wave's SECOND tile on passed every test for two rounds, because the suite's segments are small
and the grid is large.  This table is run in two regimes by tests/test_gpu_kernel_coverage.py, each under
`rocprofv3 --kernel-trace`:
    tiny    100 003 docs with every grid sized for ONE compute unit (PINOT_GPU_TEST_CUS=1): ~50 tiles for ~20 waves
    large   12 300 017 docs on the full grid: more tiles than resident waves for every kernel of the table
Every entry is compared with the oracle in this process (and its pg_result.dominant_kernel with the family it names); the test then
reads the kernel trace and requires that EVERY kernel the library contains (the device stubs of libpinot_gpu.so, `nm -C`) was
dispatched -- a kernel added without an entry here fails it.  Nothing here is product code.

    python tools/kernel_coverage.py --regime tiny|large [--only REGEX] [--list]
Prints one JSON line: {"regime", "entries", "failed": [...], "seconds"}; exit status 1 when an entry failed.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

SIZES = {"tiny": 100_003, "large": 12_300_017}

# column numbers of the table's segment
V, F, K, W8, W16, W32, A, B, C2, RI, RL, RD, DL, X, Y, Z, K1, K2, K3, WN, K4 = range(21)


def irregular(card, seed, lo=-2 ** 31, hi=2 ** 31 - 1):
    rng = np.random.default_rng(seed)
    vals = np.unique(rng.integers(lo, hi, int(card * 1.1) + 64, dtype=np.int64))
    pick = np.sort(rng.permutation(vals.shape[0])[:card])
    return vals[pick].astype(np.int32)


def build_segment(S, n, seed=1):
    ids = lambda s, card: S.synthetic_dict_ids(seed * 1000 + s, 0, n, card)
    rng = np.random.default_rng(seed)
    cols = [None] * 21
    cols[V] = S.Column.synthetic_uniform("v", n, (np.arange(100000, dtype=np.int64) * 7 + 3).astype(np.int32), seed=seed * 1000 + 1)
    cols[F] = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=seed * 1000 + 2)
    cols[K] = S.Column.synthetic_uniform("k", n, np.arange(1000, dtype=np.int32) * 3 - 7, seed=seed * 1000 + 3)
    cols[W8] = S.Column.from_dict_ids("w8", irregular(100000, 11), ids(4, 100000))
    cols[W16] = S.Column.from_dict_ids("w16", irregular(60000, 12), ids(5, 60000))
    cols[W32] = S.Column.from_dict_ids("w32", irregular(30000, 13), ids(6, 30000))
    cols[A] = S.Column.from_dict_ids("a", np.arange(200, dtype=np.int32) * 2 - 100, ids(7, 200))
    cols[B] = S.Column.from_dict_ids("b", np.arange(13, dtype=np.int32) * 5, ids(8, 13))
    cols[C2] = S.Column.from_dict_ids("c", np.arange(3, dtype=np.int32), ids(9, 3))
    cols[RI] = S.Column.raw("ri", ids(10, 1_000_000) - 500_000)
    cols[RL] = S.Column.raw_typed("rl", (ids(11, 1 << 20).astype(np.int64) - (1 << 19)) * ((1 << 20) + 3))
    cols[RD] = S.Column.raw_typed("rd", (ids(12, 1 << 20).astype(np.float64) - (1 << 19)) * 0.37)
    dl_vals = (np.arange(500, dtype=np.int64) - 250) * (2 ** 33 + 7)
    dl_ids = ids(13, 500)
    dl = S.Column.dict_encoded_typed("dl", dl_vals[dl_ids])
    cols[DL] = dl
    cols[X] = S.Column.from_dict_ids("x", np.arange(50, dtype=np.int32), ids(14, 50), with_inverted=True)
    cols[Y] = S.Column.from_dict_ids("y", np.arange(40, dtype=np.int32), ids(15, 40), with_inverted=True)
    cols[Z] = S.Column.from_dict_ids("z", np.arange(30, dtype=np.int32), ids(16, 30), with_inverted=True)
    cols[K1] = S.Column.synthetic_uniform("k1", n, np.arange(3000, dtype=np.int32), seed=seed * 1000 + 17)
    cols[K2] = S.Column.synthetic_uniform("k2", n, np.arange(2500, dtype=np.int32), seed=seed * 1000 + 18)
    cols[K3] = S.Column.synthetic_uniform("k3", n, np.arange(700, dtype=np.int32), seed=seed * 1000 + 19)
    cols[WN] = S.Column.from_dict_ids("wn", irregular(100000, 14, 0, 1 << 20), ids(20, 100000))
    cols[K4] = S.Column.synthetic_uniform("k4", n, np.arange(40000, dtype=np.int32), seed=seed * 1000 + 21)
    del rng
    return S.SegmentData("coverage_%d" % n, n, cols)


def table(Q, n):
    """[(id, {env}, family, spec)] -- `family` is what pg_result.dominant_kernel must say (None: not checked)."""
    L = Q.leaf
    f_lt = lambda t: L(Q.Pred.dict_range(F, 0, t))
    a_lt = lambda t: L(Q.Pred.dict_range(A, 0, t))
    inv = lambda col, lo, hi: L(Q.Pred.dict_range(col, lo, hi, inverted=True))
    and3_inv = Q.and_(inv(X, 3, 9), inv(Y, 0, 11), inv(Z, 5, 20))
    T = []
    add = lambda *e: T.append(e)
    # ---- the lane-private scan kernels ----
    add("simple", {}, "scan_simple_kernel", Q.QuerySpec([(Q.SUM, V)], filter=f_lt(100)))
    add("simple-set", {}, "scan_simple_kernel", Q.QuerySpec([(Q.SUM, V), (Q.COUNT, -1)], filter=L(Q.Pred.dict_set(F, list(range(0, 300, 3)), 1000))))      # scan_simple_set_kernel: the one leaf an IN list, its words in LDS
    add("raw", {}, "scan_raw_kernel", Q.QuerySpec([(Q.COUNT, -1)], filter=L(Q.Pred.raw_range(RI, -1000, 250000))))
    add("raw-sum", {}, "scan_raw_kernel", Q.QuerySpec([(Q.SUM, RI), (Q.MAX, RI)], filter=L(Q.Pred.raw_range(RI, -1000, 250000))))
    add("private-1", {}, "scan_private_kernel", Q.QuerySpec([(Q.SUM, V), (Q.COUNT, -1)], filter=Q.and_(f_lt(300), L(Q.Pred.dict_range(K, 100, 900)))))
    add("private-4", {}, "scan_private_kernel", Q.QuerySpec([(Q.SUM, V), (Q.MAX, F), (Q.MIN, K)], filter=Q.or_(f_lt(100), L(Q.Pred.dict_set(K, [1, 5, 77, 500, 999], 1000)))))
    add("typed-1", {}, "scan_private_typed_kernel", Q.QuerySpec([(Q.SUM, RL), (Q.MIN, RL)], filter=Q.and_(f_lt(400), a_lt(150))))
    add("typed-2", {}, "scan_private_typed_kernel", Q.QuerySpec([(Q.SUM, RD), (Q.MAX, DL)], filter=f_lt(500)))
    add("typed-4", {}, "scan_private_typed_kernel", Q.QuerySpec([(Q.SUM, RL), (Q.MIN, RD), (Q.SUM, DL), (Q.AVG, RD)], filter=f_lt(500)))
    add("hist-8", {}, "scan_hist_kernel", Q.QuerySpec([(Q.SUM, W8)], filter=f_lt(100)))
    add("hist-16", {}, "scan_hist_kernel", Q.QuerySpec([(Q.SUM, W16), (Q.MAX, W16)], filter=f_lt(100)))
    add("hist-32", {}, "scan_hist_kernel", Q.QuerySpec([(Q.SUM, W32)], filter=f_lt(700)))
    add("hist-8-fused-range", {}, "scan_hist_kernel", Q.QuerySpec([(Q.SUM, W8)], filter=L(Q.Pred.dict_range(W8, 30000, 60000))))
    add("hist-8-guarded", {"PINOT_GPU_HIST_GUARD": "1"}, "scan_hist_kernel", Q.QuerySpec([(Q.SUM, W8)], filter=f_lt(100)))
    add("hist-16-guarded", {"PINOT_GPU_HIST_GUARD": "1"}, "scan_hist_kernel", Q.QuerySpec([(Q.SUM, W16)], filter=f_lt(100)))
    add("narrow-single", {}, "scan_narrow_kernel", Q.QuerySpec([(Q.COUNT, -1)], filter=a_lt(77)))
    add("narrow-tree", {}, "scan_narrow_kernel", Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(a_lt(150), Q.or_(L(Q.Pred.dict_range(B, 0, 4)), L(Q.Pred.dict_range(C2, 1, 2))))))
    add("plane", {}, None, Q.QuerySpec([(Q.SUM, WN)], filter=f_lt(100)))                      # materialize_plane_kernel on first use (PINOT_GPU_HIST=0 below)
    add("plane-built", {"PINOT_GPU_HIST": "0", "PINOT_GPU_PLANE_ASYNC": "0"}, None, Q.QuerySpec([(Q.SUM, WN), (Q.MIN, V)], filter=f_lt(100)))
    add("wide-plane", {"PINOT_GPU_WIDE_PLANE": "1", "PINOT_GPU_PLANE_ASYNC": "0"}, None, Q.QuerySpec([(Q.SUM, DL)]))
    # ---- index-led ----
    add("sparse-1", {}, None, Q.QuerySpec([(Q.SUM, V)], filter=and3_inv))
    add("sparse-4", {}, None, Q.QuerySpec([(Q.SUM, V), (Q.MAX, F), (Q.MIN, K)], filter=and3_inv))
    eq3 = Q.and_(inv(X, 3, 4), inv(Y, 5, 6), inv(Z, 7, 8))                        # ~1 survivor per window: aggregated inside index_and_kernel
    add("index-gather-1", {}, "index_and_kernel", Q.QuerySpec([(Q.SUM, V), (Q.COUNT, -1)], filter=eq3))
    add("index-gather-2", {}, "index_and_kernel", Q.QuerySpec([(Q.SUM, V), (Q.MAX, F), (Q.MIN, V)], filter=eq3))
    add("index-count", {}, None, Q.QuerySpec([(Q.COUNT, -1)], filter=and3_inv))
    add("index-or-scan", {}, None, Q.QuerySpec([(Q.SUM, V)], filter=Q.or_(inv(X, 3, 5), f_lt(20))))
    add("index-and-scan", {}, None, Q.QuerySpec([(Q.SUM, V), (Q.SUM, RL)], filter=Q.and_(inv(X, 3, 20), inv(Y, 0, 30), f_lt(500))))
    add("index-not", {}, None, Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, V)], filter=Q.and_(Q.not_(inv(X, 3, 20)), inv(Y, 0, 3))))
    add("index-empty-leaf", {}, None, Q.QuerySpec([(Q.COUNT, -1), (Q.SUM, V)], filter=Q.or_(L(Q.Pred.dict_set(X, [], 50, inverted=True)), inv(Y, 3, 5))))   # fill_words_kernel
    add("index-group", {}, None, Q.QuerySpec([(Q.SUM, V)], filter=and3_inv, group_by=[K]))
    # ---- the LDS-staged kernels (fallbacks) ----
    staged = {"PINOT_GPU_SCAN_PRIVATE": "0"}
    add("staged-1", staged, "scan_agg_kernel", Q.QuerySpec([(Q.SUM, V)], filter=f_lt(100)))
    add("staged-4", staged, "scan_agg_kernel", Q.QuerySpec([(Q.SUM, V), (Q.MAX, F), (Q.MIN, K)], filter=f_lt(100)))
    add("staged-typed", dict(staged, PINOT_GPU_SCAN_TYPED_PRIVATE="0"), "scan_agg_kernel", Q.QuerySpec([(Q.SUM, RL), (Q.MAX, DL)], filter=f_lt(100)))
    nodma = dict(staged, PINOT_GPU_NO_DMA="1")
    add("staged-1-nodma", nodma, "scan_agg_kernel", Q.QuerySpec([(Q.SUM, V)], filter=f_lt(100)))
    add("staged-4-nodma", nodma, "scan_agg_kernel", Q.QuerySpec([(Q.SUM, V), (Q.MAX, F), (Q.MIN, K)], filter=f_lt(100)))
    add("staged-typed-nodma", dict(nodma, PINOT_GPU_SCAN_TYPED_PRIVATE="0"), "scan_agg_kernel", Q.QuerySpec([(Q.SUM, RL), (Q.MAX, DL)], filter=f_lt(100)))
    add("finalize-launch", {"PINOT_GPU_FOLD_FINALIZE": "0"}, "scan_simple_kernel", Q.QuerySpec([(Q.SUM, V)], filter=f_lt(100)))
    # ---- group-by ----
    add("group-lds", {}, "group_private_kernel", Q.QuerySpec([(Q.SUM, V), (Q.MAX, F)], group_by=[K]))
    add("group-lds-filter", {}, "group_private_kernel", Q.QuerySpec([(Q.SUM, V), (Q.COUNT, -1), (Q.MIN, F)], filter=f_lt(300), group_by=[B, C2]))
    nopart = {"PINOT_GPU_GROUP_PARTITION": "0"}
    add("group-direct", nopart, "group_private_kernel", Q.QuerySpec([(Q.SUM, V)], filter=f_lt(500), group_by=[K, K3]))                 # 700 000 slots, HBM atomics
    add("group-direct-wide", nopart, "group_private_kernel", Q.QuerySpec([(Q.COUNT, -1)], filter=f_lt(30), group_by=[K1, K4]))         # 1.2e8 slots > 2^24
    add("group-hash-long", {}, "group_private_kernel", Q.QuerySpec([(Q.SUM, V)], filter=f_lt(20), group_by=[K1, K2, K4]))              # 3e11 raw keys: LongMap holder
    add("group-hash-array", {}, "group_private_kernel", Q.QuerySpec([(Q.COUNT, -1)], filter=f_lt(20), group_by=[K1, K2, K4, V, W8]))   # beyond a long: ArrayMap holder
    add("group-limit", {}, None, Q.QuerySpec([(Q.SUM, V)], group_by=[K, K3], num_groups_limit=1000))                                   # group_first_doc_kernel
    force = {"PINOT_GPU_GROUP_PARTITION": "force"}
    add("partition-count", force, "group_partition_scatter_kernel", Q.QuerySpec([(Q.COUNT, -1)], group_by=[K, K3]))
    add("partition-1-packed", force, "group_partition_scatter_kernel", Q.QuerySpec([(Q.MAX, F)], filter=f_lt(800), group_by=[K, K3]))
    unpacked = dict(force, PINOT_GPU_PARTITION_PACKED="0")
    add("partition-0", unpacked, "group_partition_scatter_kernel", Q.QuerySpec([(Q.COUNT, -1)], group_by=[K, K3]))
    add("partition-1", unpacked, "group_partition_scatter_kernel", Q.QuerySpec([(Q.SUM, V)], group_by=[K, K3]))
    add("partition-2", force, "group_partition_scatter_kernel", Q.QuerySpec([(Q.SUM, V), (Q.MAX, F)], filter=f_lt(900), group_by=[K, K3]))
    add("partition-3", force, "group_partition_scatter_kernel", Q.QuerySpec([(Q.SUM, V), (Q.MAX, F), (Q.MIN, A)], group_by=[K, K3]))
    add("partition-two-level", force, "group_partition_scatter_kernel", Q.QuerySpec([(Q.SUM, V)], group_by=[K1, K2]))                  # 7.5 M raw keys
    add("group-typed", {}, None, Q.QuerySpec([(Q.SUM, RL), (Q.MAX, RD)], filter=f_lt(500), group_by=[K]))
    add("group-typed-wide", {}, None, Q.QuerySpec([(Q.SUM, RL)], filter=f_lt(10), group_by=[K1, K4]))
    add("group-typed-hash", {}, None, Q.QuerySpec([(Q.SUM, RL), (Q.MIN, RD)], filter=f_lt(20), group_by=[K1, K2, K4]))                 # raw 8-byte inputs under a LongMap holder: group_typed_direct_kernel<false, true>
    add("group-double-key", {}, None, Q.QuerySpec([(Q.SUM, V), (Q.COUNT, -1)], filter=f_lt(500), group_by=[RD], num_groups_limit=1000))           # rank_image_keys / rank_image_pack kernels (+ rocPRIM)
    add("group-wide-long-key", {}, None, Q.QuerySpec([(Q.MAX, F)], filter=f_lt(50), group_by=[RL, B], num_groups_limit=500))
    add("group-raw-key", {}, None, Q.QuerySpec([(Q.SUM, V)], filter=L(Q.Pred.raw_range(RI, 0, 5000)), group_by=[RI]))                 # raw_min_max / build_raw_key_image
    gstaged = {"PINOT_GPU_GROUP_PRIVATE": "0", "PINOT_GPU_GROUP_PARTITION": "0"}
    add("group-staged-lds", gstaged, "scan_group_kernel", Q.QuerySpec([(Q.SUM, V), (Q.MAX, F)], filter=f_lt(500), group_by=[K]))
    add("group-staged-hbm", gstaged, "scan_group_kernel", Q.QuerySpec([(Q.SUM, V)], filter=f_lt(500), group_by=[K, K3]))
    add("group-staged-wide", gstaged, "scan_group_kernel", Q.QuerySpec([(Q.COUNT, -1)], filter=f_lt(30), group_by=[K1, K4]))
    gnodma = dict(gstaged, PINOT_GPU_NO_DMA="1")
    add("group-staged-lds-nodma", gnodma, "scan_group_kernel", Q.QuerySpec([(Q.SUM, V), (Q.MAX, F)], filter=f_lt(500), group_by=[K]))
    add("group-staged-hbm-nodma", gnodma, "scan_group_kernel", Q.QuerySpec([(Q.SUM, V)], filter=f_lt(500), group_by=[K, K3]))
    add("group-staged-wide-nodma", gnodma, "scan_group_kernel", Q.QuerySpec([(Q.COUNT, -1)], filter=f_lt(30), group_by=[K1, K4]))
    # ---- numEntriesScannedInFilter on the device ----
    and3 = Q.and_(f_lt(300), L(Q.Pred.dict_range(K, 0, 500)), L(Q.Pred.dict_range(K3, 100, 600)))
    add("fsm-fused-1", {}, "scan_private_kernel", Q.QuerySpec([(Q.SUM, V)], filter=and3))                                    # scan_private_fsm_kernel<1>: the walk inside the scan
    add("fsm-fused-4", {}, "scan_private_kernel", Q.QuerySpec([(Q.SUM, V), (Q.MAX, F), (Q.MIN, K)], filter=Q.and_(f_lt(300), Q.or_(L(Q.Pred.dict_range(K, 0, 100)), L(Q.Pred.dict_range(K3, 0, 50))))))
    add("fsm-pass-behind-the-scan", {"PINOT_GPU_FSM_FUSED": "0"}, "scan_private_kernel", Q.QuerySpec([(Q.SUM, V)], filter=and3))
    add("leap2", {"PINOT_GPU_FSM_STATS": "0", "PINOT_GPU_EXACT_FILTER_STATS_DOCS": "0"}, None, Q.QuerySpec([(Q.SUM, V)], filter=Q.and_(f_lt(100), a_lt(50))))
    return T


def fsm_trees(Q, rng, n, count):
    """Root ANDs of 2..8 children over the scan / index columns: the transducer's machines (2..16 states, 2..8 inputs)."""
    L = Q.leaf

    def scan_leaf():
        k = int(rng.integers(0, 4))
        if k == 0:
            lo = int(rng.integers(0, 150)); return L(Q.Pred.dict_range(A, lo, lo + int(rng.integers(1, 50)), exclusive=bool(rng.integers(0, 2))))
        if k == 1:
            return L(Q.Pred.dict_range(C2, int(rng.integers(0, 2)), int(rng.integers(2, 4))))
        if k == 2:
            lo = int(rng.integers(0, 9)); return L(Q.Pred.dict_range(B, lo, lo + int(rng.integers(1, 5))))
        return L(Q.Pred.dict_set(A, sorted(set(int(x) for x in rng.integers(0, 200, size=40))), 200, exclusive=bool(rng.integers(0, 2))))

    def index_leaf():
        k = int(rng.integers(0, 3))
        if k == 0:
            return L(Q.Pred.dict_range(X, int(rng.integers(0, 25)), 50, inverted=True, exclusive=bool(rng.integers(0, 2))))
        if k == 1:
            return L(Q.Pred.dict_set(Y, sorted(set(int(x) for x in rng.integers(0, 40, size=12))), 40, inverted=True))
        lo = int(rng.integers(0, n)); return L(Q.Pred.doc_range(lo, min(n - 1, lo + int(rng.integers(0, n)))))

    out = []
    # NOT children over scan leaves: the episode kernels (fsm_chunk_states / fsm_tile_states / fsm_episode_tiles / fsm_episode_finish), with a
    # leaf that matches rarely (episodes of many batches), one that matches half the docs, an index-based leader, an OR beside the NOT
    rare, half, some = L(Q.Pred.dict_range(A, 7, 8)), L(Q.Pred.dict_range(C2, 0, 1)), L(Q.Pred.dict_range(B, 2, 5))
    posting = L(Q.Pred.dict_range(X, 0, 20, inverted=True))
    out += [Q.and_(some, Q.not_(rare)), Q.and_(Q.not_(half), some), Q.and_(posting, Q.not_(rare)), Q.and_(some, Q.not_(half), Q.or_(rare, posting)),
            Q.and_(posting, some, Q.not_(L(Q.Pred.dict_range(A, 0, 150)))), Q.and_(some, Q.not_(posting), Q.not_(rare)),
            # NOT over an OR of leaves: an episode stream per scan member; nine to sixteen states over three / four inputs:
            # fsm_tile_fns16_kernel<3 | 4> + fsm_episode_ranges_kernel<16, 4> (with PINOT_GPU_FSM_PERM=0: the table walks)
            Q.and_(some, Q.not_(Q.or_(half, rare))), Q.and_(Q.not_(Q.or_(rare, some)), half), Q.and_(posting, some, Q.not_(Q.or_(half, rare))),
            Q.and_(some, Q.not_(Q.or_(posting, rare)))][:max(count, 0)]
    # machines of the rarer (states, inputs) classes, found by a search over shapes (tools/fstats/fstats_driver.cpp fstats_fsm_class): a child
    # is a pool index or an OR of pool indexes; nine to thirteen states over three inputs, four to fifteen over four
    frozen = [(3, [[2, 1, 0], 1, [2, 1, 0]]), (3, [[2, 0], [1, 2], 0, [2, 1]]), (3, [[0, 2, 1], [1, 1], [0, 2, 2], [0, 2]]), (3, [[1, 0], 2, 0, [0, 1, 2]]),
              (4, [[1, 0, 2], [2, 1], 3, 0]), (4, [[1, 0], 2, [1, 3], 3, 1]), (4, [0, 0, 2, [1, 3, 2]]), (2, [[0, 0, 1], [0, 0, 1]]), (2, [[0, 0], [0, 1], 1]),
              (3, [[2, 0, 0], [1, 2], [0, 0, 1], [1, 0], [2, 2], [2, 1]]), (4, [[3, 3, 1], [1, 1, 3], 3, 0, [0, 0], [2, 2, 1], 2])]
    for num, shape in frozen[:count]:
        pool = [L(Q.Pred.dict_range(A, 10 * i, 10 * i + 60)) if i % 3 == 0 else L(Q.Pred.dict_range(B, i % 5, i % 5 + 4)) if i % 3 == 1 else L(Q.Pred.dict_range(C2, i % 2, i % 2 + 1))
                for i in range(num)]
        out.append(Q.and_(*[pool[c] if isinstance(c, int) else Q.or_(*[pool[x] for x in c]) for c in shape]))
    # machines over FEW inputs: two to four predicates, each behind several leaves of the tree (one Pred object = one predicate)
    for t in range(count // 2):
        pool = [scan_leaf() if rng.integers(0, 4) else index_leaf() for _p in range(2 + t % 3)]
        if all(p.pred.inverted or p.pred.kind == 5 for p in pool):      # (5 = PG_PRED_DOC_RANGE)
            pool[0] = scan_leaf()                                       # at least one scan leaf: an index-only tree scans nothing
        pick = lambda: pool[int(rng.integers(0, len(pool)))]
        kids = []
        for _c in range(2 + int(rng.integers(0, 5))):
            kids.append(pick() if rng.integers(0, 3) == 0 else Q.or_(*[pick() for _m in range(int(rng.integers(2, 4)))]))
        out.append(Q.and_(*kids))
    for t in range(count - count // 2):
        kids = []
        for _c in range(2 + t % 7):
            r = int(rng.integers(0, 10))
            if r < 5:
                kids.append(scan_leaf())
            elif r < 7:
                kids.append(index_leaf())
            else:
                kids.append(Q.or_(*[scan_leaf() if rng.integers(0, 3) else index_leaf() for _m in range(int(rng.integers(2, 4)))]))
        out.append(Q.and_(*kids))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--regime", default="tiny", choices=sorted(SIZES))
    ap.add_argument("--only", default="")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--fsm-trees", type=int, default=-1, help="random root ANDs for the transducer kernels (default: 400 tiny, 40 large)")
    ap.add_argument("--flip", default="", help="NAME=value,...: switches set for the WHOLE table (an entry's own environment wins); the answers are what is "
                                               "checked then, not which kernel gave them -- the non-default side of the launch-structure switches")
    args = ap.parse_args()
    t_begin = time.time()
    import helpers as H
    from oracle import oracle
    from pinot_amd import _abi
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine

    n = SIZES[args.regime]
    entries = table(Q, n)
    flipped = dict(kv.split("=", 1) for kv in args.flip.split(",") if kv)
    os.environ.update(flipped)
    if args.list:
        for e in entries:
            print(e[0], e[1], e[2])
        return 0
    only = re.compile(args.only) if args.only else None
    if args.regime == "tiny":
        os.environ["PINOT_GPU_TEST_CUS"] = "1"
    os.environ["PINOT_GPU_PLANE_ASYNC"] = "0"
    engine = Engine(device_id=0, time_kernels=False)
    seg = build_segment(S, n)
    failed, ran = [], 0
    wants = {}

    def want_of(key, spec):
        if key not in wants:
            wants[key] = oracle.execute(seg, spec)
        return wants[key]

    def check(eid, got, want, family):
        try:
            H.assert_results_equal(got, want)
            if family is not None and not flipped:
                assert got.dominant_kernel == family, "dominant kernel %s, expected %s" % (got.dominant_kernel, family)
        except AssertionError as e:
            failed.append({"id": eid, "error": str(e)[:300]})

    # ---- the table: entries with the same environment share one pg_init and one open segment ----
    by_env = {}
    for e in entries:
        if only is None or only.search(e[0]):
            by_env.setdefault(tuple(sorted(e[1].items())), []).append(e)
    for env, group in by_env.items():
        engine.reinit(**dict(env))
        try:
            with engine.open(seg) as g:
                for eid, _env, family, spec in group:
                    status = g.check(spec)
                    if status != 0:
                        failed.append({"id": eid, "error": "pg_query_check %d" % status})
                        continue
                    got = g.execute(spec)
                    check(eid, got, want_of(eid, spec), family)
                    again = g.execute(spec)            # planes / key images / tiers are in place now: the steady-state kernel
                    check(eid + "(again)", again, want_of(eid, spec), family)
                    ran += 1
        finally:
            engine.reinit(**{k: flipped.get(k) for k, _ in env})
    # ---- the SPI readers and the diagnostics ----
    if only is None or only.search("spi"):
        with engine.open(seg) as g:
            docs = np.sort(np.random.default_rng(3).choice(n, size=4000, replace=False)).astype(np.int32)
            assert np.array_equal(g.read_dict_ids(V, docs), oracle.read_dict_ids(seg.columns[V].fwd, seg.columns[V].bits, n, docs))
            assert np.array_equal(g.read_int_values(W8, docs), oracle.read_int_values(seg, W8, docs))
            assert np.array_equal(g.read_double_values(RD, docs), oracle.read_double_values(seg, RD, docs)[0])
            bits, card = g.filter_bitmap(Q.QuerySpec([], filter=Q.leaf(Q.Pred.match_all())))
            assert card == n
        import ctypes as C
        peak = C.c_double()
        assert engine.lib.pg_measure_stream_read(0, 256 << 20, 2, C.byref(peak)) == _abi.PG_OK
        ran += 1
    # ---- null handling: a nullable group key (build_nullkey_fwd_kernel) ----
    if only is None or only.search("nulls"):
        m = min(n, 300_007)
        rng = np.random.default_rng(9)
        kv = rng.integers(0, 50, m).astype(np.int32)
        nulls = rng.random(m) < 0.1
        kv[nulls] = np.iinfo(np.int32).min
        vv = rng.integers(0, 1000, m).astype(np.int32)
        nseg = S.SegmentData("nullkeys", m, [S.Column.dict_encoded("kn", kv).with_nulls(nulls), S.Column.dict_encoded("vn", vv)])
        spec = Q.QuerySpec([(Q.SUM, 1), (Q.COUNT, -1)], group_by=[0], null_handling=True)
        with engine.open(nseg) as g:
            if g.check(spec) == 0:
                check("nulls", g.execute(spec), oracle.execute(nseg, spec), None)
        ran += 1
    # ---- the transducer's kernels: byte-function walks, then table walks of the same machines ----
    if only is None or only.search("fsm"):
        count = args.fsm_trees if args.fsm_trees >= 0 else (400 if args.regime == "tiny" else 40)
        trees = fsm_trees(Q, np.random.default_rng(77), n, count)
        for env in ({"PINOT_GPU_EXACT_FILTER_STATS_DOCS": "0"}, {"PINOT_GPU_EXACT_FILTER_STATS_DOCS": "0", "PINOT_GPU_FSM_PERM": "0"}):
            engine.reinit(**env)
            try:
                with engine.open(seg) as g:
                    for t, flt in enumerate(trees):
                        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=flt)
                        if len(spec.predicates) > 8 or spec.c.num_filter_nodes > 24 or g.check(spec) != 0:
                            continue
                        got = g.execute(spec)
                        want = want_of("fsm%d" % t, spec)
                        check("fsm%d%s" % (t, "-tables" if "PINOT_GPU_FSM_PERM" in env else ""), got, want, None)
                        if got.filter_entries_exact and got.stats[1] != want.stats[1]:
                            failed.append({"id": "fsm%d" % t, "error": "numEntriesScannedInFilter %d, oracle %d" % (got.stats[1], want.stats[1])})
                        ran += 1
            finally:
                engine.reinit(**{k: flipped.get(k) for k in env})
    # ---- pg_execute_batch: one launch per kind ----
    if only is None or "batch" in args.only:
        parts = 4 if args.regime == "tiny" else 3
        m = n // parts
        segs = [build_segment(S, m, seed=10 + i) for i in range(parts)]
        L = Q.leaf
        f_lt = lambda t: L(Q.Pred.dict_range(F, 0, t))
        shapes = {
            # (root ORs: a root AND of scan leaves is counted by the leap-frog / transducer passes and keeps a launch of its own)
            "batch-private-1": Q.QuerySpec([(Q.SUM, V), (Q.COUNT, -1)], filter=Q.or_(f_lt(300), L(Q.Pred.dict_range(K, 100, 300)))),
            # (dictId-range leaves only: a set leaf uploads its words ahead of the kernel, and an item with work of its own ahead of it is not shared)
            "batch-private-4": Q.QuerySpec([(Q.SUM, V), (Q.MAX, F), (Q.MIN, K)], filter=Q.or_(f_lt(100), L(Q.Pred.dict_range(K, 400, 460)))),
            "batch-simple": Q.QuerySpec([(Q.SUM, V)], filter=f_lt(100)),
            "batch-simple-set": Q.QuerySpec([(Q.SUM, V), (Q.COUNT, -1)], filter=L(Q.Pred.dict_set(F, list(range(0, 300, 3)), 1000))),      # scan_lean_batch_kernel<13>
            "batch-raw": Q.QuerySpec([(Q.COUNT, -1)], filter=L(Q.Pred.raw_range(RI, -1000, 250000))),
            "batch-hist-8": Q.QuerySpec([(Q.SUM, W8)], filter=f_lt(100)),
            "batch-hist-16": Q.QuerySpec([(Q.SUM, W16)], filter=f_lt(100)),
            "batch-hist-32": Q.QuerySpec([(Q.SUM, W32)], filter=f_lt(100)),
            "batch-narrow": Q.QuerySpec([(Q.COUNT, -1)], filter=Q.or_(L(Q.Pred.dict_range(A, 0, 50)), Q.not_(L(Q.Pred.dict_range(B, 0, 9))), L(Q.Pred.dict_range(C2, 1, 2)))),
            "batch-narrow-single": Q.QuerySpec([(Q.COUNT, -1)], filter=L(Q.Pred.dict_range(A, 0, 77))),
            "batch-typed-1": Q.QuerySpec([(Q.SUM, RL), (Q.MIN, RL)], filter=Q.or_(f_lt(200), L(Q.Pred.dict_range(A, 0, 50)))),
            "batch-typed-2": Q.QuerySpec([(Q.SUM, RD), (Q.MAX, DL)], filter=f_lt(500)),
            "batch-typed-4": Q.QuerySpec([(Q.SUM, RL), (Q.MIN, RD), (Q.SUM, DL), (Q.AVG, RD)], filter=f_lt(500)),
            "batch-group": Q.QuerySpec([(Q.SUM, V), (Q.MAX, F)], filter=f_lt(700), group_by=[K]),
            # index-led items whose whole device work is index_and_kernel publishing the record: one launch of index_and_batch_kernel
            "batch-index-count": Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(L(Q.Pred.dict_range(X, 3, 4, inverted=True)), L(Q.Pred.dict_range(Y, 5, 6, inverted=True)))),
            "batch-index-gather": Q.QuerySpec([(Q.SUM, V), (Q.MAX, F)], filter=Q.and_(L(Q.Pred.dict_range(X, 3, 4, inverted=True)), L(Q.Pred.dict_range(Y, 5, 6, inverted=True)),
                                                                                     L(Q.Pred.dict_range(Z, 7, 8, inverted=True)))),
        }
        opened = [engine.open(s) for s in segs]
        try:
            for sid, spec in shapes.items():
                if only is not None and not only.search(sid):
                    continue
                for rep in range(2):
                    for i, (status, res) in enumerate(engine.execute_batch(opened, [spec] * parts)):
                        if status != _abi.PG_OK:
                            failed.append({"id": "%s[%d]" % (sid, i), "error": "status %d" % status})
                            continue
                        check("%s[%d]" % (sid, i), res, want_of("%s[%d]" % (sid, i), spec) if False else oracle.execute(segs[i], spec), None)
                ran += 1
        finally:
            [g.close() for g in opened]
    print(json.dumps({"regime": args.regime, "docs": n, "entries": ran, "failed": failed, "seconds": round(time.time() - t_begin, 1)}))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
