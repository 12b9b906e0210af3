"""The `variants` array of bench.py's JSON line: every BASELINE.json configuration besides the headline, timed in the driver's own
run with the parity check on (rank 0, N = 1).

Each entry: {id, config, query, rows, kernel, kernel_ms (the dominant kernel, HIP events on the launch stream), all_kernels_ms (every
kernel of the query, e.g. posting expansion), algorithmic_bytes (SURVEY.md 8(d) / BASELINE.md section 3), achieved_GBps and frac of
8 TB/s on all_kernels_ms, rows_per_s, bit_exact_vs_oracle}.  The oracle runs on all host cores (oracle.execute_sliced) so that a
1 B-row check takes a fraction of a second; inverted-index leaves are checked against the oracle's SCAN of the same predicate (the
same docId set by definition; the oracle's own posting reader is pinned in tests/).
"""
import os
import time

import numpy as np

HBM_PEAK_GBPS = 8000.0


def _shared(S, col, name, dict_values):
    """A column with the same packed dictIds as `col` (the host buffer is shared) under another dictionary."""
    lib = S.load_host_library()
    dict_values = np.ascontiguousarray(dict_values, dtype=np.int32)
    dictionary = np.zeros(dict_values.shape[0] * 4, dtype=np.uint8)
    lib.ph_dict_write_int(S._i32p(dict_values), int(dict_values.shape[0]), S._u8p(dictionary))
    return S.Column(name, col.encoding, col.bits, col.cardinality, col.fwd, dictionary, None, dict_values)


def run(engine, gseg0, seg0, n, n_c5, match, check=True, steps=10, warmup=40):
    from bench import Timer, v_dictionary
    from oracle import oracle
    from pinot_amd import _abi
    from pinot_amd import query as Q
    from pinot_amd import segment as S

    # warmup = 40 untimed launches per variant: the GPU idles while the host generates columns and runs the oracle, and the first
    # launches after an idle period run through the clock transient (same reason as bench.py's clock settle)
    timer = Timer(engine.lib, _abi)
    out = []
    want = lambda vid: match is None or match.search(vid) is not None

    def cpu_port(seg, spec, sample_rows):
        """A 1-core cpu_baseline for a variant: the C oracle ("port": no JDK here) over a bounded prefix of the same workload."""
        res, secs, ran = oracle.execute_prefix(seg, spec, sample_rows)
        return {"value": ran / secs, "unit": "rows/s", "cores": 1, "kind": "port",
                "sample": "the first %d of %d rows of the same segment and query through the C oracle on one host core; %.2f s" % (ran, seg.num_docs, secs)}

    def report(vid, config, query, rows, nbytes, gseg, seg, spec, oracle_spec=None, extra=None, cpu_rows=0):
        t = timer.run(gseg, spec, steps, warmup)
        rec = {"id": vid, "config": config, "query": query, "rows": rows}
        rec.update(t)
        got = gseg.execute(spec)
        if callable(nbytes):
            nbytes = nbytes(got.stats[0])                    # bytes that depend on how many docs matched
        ms = t["all_kernels_ms"] if t["all_kernels_ms"] > 0 else float("inf")      # metadata-only answers launch nothing
        rec.update({"algorithmic_bytes": int(nbytes), "achieved_GBps": nbytes / ms / 1e6, "frac": nbytes / ms / 1e6 / HBM_PEAK_GBPS,
                    "frac_dominant_kernel": (nbytes / t["kernel_ms"] / 1e6 / HBM_PEAK_GBPS) if t["kernel_ms"] > 0 else None,
                    "rows_per_s": rows / ms * 1e3})
        rec["docs_matched"] = got.stats[0]
        rec["bit_exact_vs_oracle"] = None
        if check:
            t0 = time.perf_counter()
            wanted = oracle.execute_sliced(seg, oracle_spec or spec)
            rec["bit_exact_vs_oracle"] = bool(oracle.matches_sliced(got, wanted, [f for f, _ in spec.aggregations]) and got.stats[0] == wanted["docs_scanned"])
            rec["oracle_check_s"] = time.perf_counter() - t0
            if cpu_rows:
                rec["cpu_baseline"] = cpu_port(seg, spec, cpu_rows)          # (the query as the reference would run it: inverted leaves through the postings)
        rec["frac_host_clock"] = (nbytes / t["step_ms_host_clock"] / 1e6 / HBM_PEAK_GBPS) if t["step_ms_host_clock"] > 0 else None
        if extra:
            rec.update(extra)
        out.append(rec)

    B = lambda col: col.fwd.nbytes
    v, f = seg0.columns[0], seg0.columns[1]

    def summed(col, rows):
        """a summed column is charged in full from 1/16 selectivity up; below that, min(B(col), M * 64 B) -- one
        64-byte sector per matching doc -- because only the matches' values are needed (the reference reads only surviving docs)."""
        return lambda matched: B(col) if matched * 16 >= rows else min(B(col), matched * 64)

    # ---- C2 / C3 on 1 B rows: the headline's v and f, v under two dictionaries without structure, and the C3 columns ----
    if any(want(x) for x in ("C2b-irregular", "C2b-window", "C2a-affine", "C2a-irregular", "C3", "C3-filter", "C3-irregular", "COUNT-filter", "C2b-1pct", "C2b-50pct", "AND3-scan", "AND-OR-scan", "AND-NOT-scan", "NOT-NOT-scan", "AND-NOT-OR-scan", "AND-NOT-OR-scan-bound", "AND3-scan-bound", "AND-OR-scan-bound", "AND-NOT-scan-bound", "NOT-NOT-scan-bound", "C2b-in-list", "C2b-irregular-in-list", "C3-in-list")):
        t0 = time.time()
        v_irr = _shared(S, v, "v_irr", v_dictionary("irregular"))
        v_win = _shared(S, v, "v_win", v_dictionary("window"))
        k = S.Column.synthetic_uniform("k", n, np.arange(1000, dtype=np.int32) * 3, seed=3)
        a = S.Column.synthetic_uniform("a", n, (np.arange(100000, dtype=np.int64) * 5 + 1).astype(np.int32), seed=4)
        b = S.Column.synthetic_uniform("b", n, np.arange(65536, dtype=np.int32) * 2, seed=5)
        a_irr = _shared(S, a, "a_irr", v_dictionary("irregular", seed=77))
        seg = S.SegmentData("variants", n, [v, f, v_irr, v_win, k, a, b, a_irr])
        gen_s = time.time() - t0
        flt = Q.leaf(Q.Pred.dict_range(1, 0, 100))
        with engine.open(seg) as g:
            setup = {"host_generate_s": gen_s, "device_bytes": g.device_bytes()}
            for vid, ci, name in (("C2b-irregular", 2, "100000 sorted distinct values from the whole int32 range"), ("C2b-window", 3, "100000 sorted distinct values from [0, 2^20)")):
                if want(vid):
                    report(vid, "BASELINE.json configs[1], dictionary without structure", "SELECT SUM(%s) WHERE f < 100 (10%%); dictionary: %s" % (seg.columns[ci].name, name), n, B(v) + B(f), g, seg,
                           Q.QuerySpec([(Q.SUM, ci)], filter=flt), extra={"dictionary": "irregular" if ci == 2 else "window", "hbm_resident_bytes_of_the_summed_column": B(v) + 400000})
            for vid, t in (("C2b-1pct", 10), ("C2b-50pct", 500)):
                if want(vid):
                    charge = summed(v, n)
                    report(vid, "BASELINE.json configs[1], other selectivities", "SELECT SUM(v) WHERE f < %d" % t, n, lambda m, charge=charge: B(f) + charge(m), g, seg,
                           Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, t))),
                           extra={"dictionary": "affine", "algorithmic_bytes_note": "B(f) + (B(v) from 1/16 selectivity up, else min(B(v), matches x 64 B)): SURVEY.md 8(d)"})
            if want("C2b-in-list"):
                # InPredicateEvaluator over f's dictIds: a dictId-set leaf (the words staged in LDS once per workgroup: pg_kernels.h stage_filter_sets)
                report("C2b-in-list", "BASELINE.json configs[1] with an IN list for a filter", "SELECT SUM(v) WHERE f IN (100 of f's 1000 values: every third of the first 300) (10%)", n, B(v) + B(f), g, seg,
                       Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_set(1, list(range(0, 300, 3)), 1000))), extra={"dictionary": "affine"})
            if want("C2b-irregular-in-list"):
                report("C2b-irregular-in-list", "BASELINE.json configs[1] with an IN list for a filter, dictionary without structure", "SELECT SUM(v_irr) WHERE f IN (100 of f's 1000 values) (10%)", n, B(v) + B(f), g, seg,
                       Q.QuerySpec([(Q.SUM, 2)], filter=Q.leaf(Q.Pred.dict_set(1, list(range(0, 300, 3)), 1000))), extra={"dictionary": "irregular"})
            for vid, ci in (("C2a-affine", 0), ("C2a-irregular", 2)):
                if want(vid):
                    report(vid, "BASELINE.md C2a (predicate on the summed column)", "SELECT SUM(%s) WHERE %s BETWEEN dict[45000] AND dict[54999] (10%%)" % (seg.columns[ci].name, seg.columns[ci].name),
                           n, B(v), g, seg, Q.QuerySpec([(Q.SUM, ci)], filter=Q.leaf(Q.Pred.dict_range(ci, 45000, 55000))), extra={"dictionary": "affine" if ci == 0 else "irregular"})
            if want("C3"):
                report("C3", "BASELINE.json configs[2]", "SELECT SUM(a), MAX(b) GROUP BY k (1000 groups)", n, B(k) + B(a) + B(b), g, seg, Q.QuerySpec([(Q.SUM, 5), (Q.MAX, 6)], group_by=[4]),
                       cpu_rows=200_000_000)
            if want("C3-filter"):
                report("C3-filter", "BASELINE.json configs[2] + filter", "SELECT SUM(a), MAX(b) WHERE f < 100 GROUP BY k", n, B(k) + B(a) + B(b) + B(f), g, seg,
                       Q.QuerySpec([(Q.SUM, 5), (Q.MAX, 6)], filter=flt, group_by=[4]))
            if want("C3-in-list"):
                report("C3-in-list", "BASELINE.json configs[2] + an IN list for a filter", "SELECT SUM(a), MAX(b) WHERE f IN (100 of f's 1000 values) GROUP BY k", n, B(k) + B(a) + B(b) + B(f), g, seg,
                       Q.QuerySpec([(Q.SUM, 5), (Q.MAX, 6)], filter=Q.leaf(Q.Pred.dict_set(1, list(range(0, 300, 3)), 1000)), group_by=[4]))
            if want("C3-irregular"):
                report("C3-irregular", "BASELINE.json configs[2], summed column with a dictionary without structure", "SELECT SUM(a_irr), MAX(b) GROUP BY k", n, B(k) + B(a) + B(b), g, seg,
                       Q.QuerySpec([(Q.SUM, 7), (Q.MAX, 6)], group_by=[4]), extra={"dictionary": "irregular"})
            # leap-frogging filters whose numEntriesScannedInFilter the device counts with the transducer pass (pg_filter_fsm.h): all_kernels_ms
            # includes that pass (the leaves' bitmaps + fsm_*_kernel, on the host clock)
            for vid, sql, flt3 in (("AND3-scan", "SELECT SUM(v) WHERE f < 300 AND k < 1500 AND b < 60000 (three scan leaves: 30% / 50% / 46%)",
                                    Q.and_(Q.leaf(Q.Pred.dict_range(1, 0, 300)), Q.leaf(Q.Pred.dict_range(4, 0, 500)), Q.leaf(Q.Pred.dict_range(6, 0, 30000)))),
                                   ("AND-OR-scan", "SELECT SUM(v) WHERE f < 300 AND (k < 300 OR b < 6000) (a scan leaf AND an OR of two)",
                                    Q.and_(Q.leaf(Q.Pred.dict_range(1, 0, 300)), Q.or_(Q.leaf(Q.Pred.dict_range(4, 0, 100)), Q.leaf(Q.Pred.dict_range(6, 0, 3000))))),
                                   # (a NOT child pulls its leaf in 256-doc batches: the episodes of pg_fsm_kernels.h, a second walk of the docs behind the count)
                                   ("AND-NOT-scan", "SELECT SUM(v) WHERE f < 300 AND NOT (k < 1500) (a scan leaf AND a NOT over a scan leaf: NotDocIdIterator)",
                                    Q.and_(Q.leaf(Q.Pred.dict_range(1, 0, 300)), Q.not_(Q.leaf(Q.Pred.dict_range(4, 0, 500))))),
                                   # (two NOT children: two episode streams of one seven-state machine, the episode pass once per stream)
                                   ("NOT-NOT-scan", "SELECT SUM(v) WHERE NOT (f < 300) AND NOT (k < 1500) (two NOTs over scan leaves: two NotDocIdIterators leap-frogging)",
                                    Q.and_(Q.not_(Q.leaf(Q.Pred.dict_range(1, 0, 300))), Q.not_(Q.leaf(Q.Pred.dict_range(4, 0, 500))))),
                                   # (NOT over an OR of two scan leaves: an episode stream per member of the OR -- OrFilterOperator.getFalses; round 6c)
                                   ("AND-NOT-OR-scan", "SELECT SUM(v) WHERE f < 300 AND NOT (k < 300 OR b < 6000) (a scan leaf AND a NOT over an OR of two scan leaves)",
                                    Q.and_(Q.leaf(Q.Pred.dict_range(1, 0, 300)), Q.not_(Q.or_(Q.leaf(Q.Pred.dict_range(4, 0, 100)), Q.leaf(Q.Pred.dict_range(6, 0, 3000))))))):
                if want(vid):
                    sp3 = Q.QuerySpec([(Q.SUM, 0)], filter=flt3)
                    report(vid, "numEntriesScannedInFilter of a leap-frogging filter at 1 B rows", sql, n, B(v) + B(f) + B(k) + (B(b) if vid not in ("AND-NOT-scan", "NOT-NOT-scan") else 0), g, seg, sp3)
                    r3 = g.execute(sp3)
                    out[-1]["filter_entries_exact"] = bool(r3.filter_entries_exact)
                    out[-1]["num_entries_scanned_in_filter"] = int(r3.stats[1])
                    if check:
                        # (the oracle replays the iterator objects doc by doc on one core: ~75 s per 1 B docs for the NOT filter, done for that variant only)
                        if n <= 200_000_000 or (vid in ("AND-NOT-scan", "NOT-NOT-scan", "AND-NOT-OR-scan") and os.environ.get("PINOT_BENCH_CHECK_ENTRIES_1B") == "1"):
                            t_o = time.perf_counter()
                            out[-1]["oracle_num_entries_scanned_in_filter"] = int(oracle.execute(seg, Q.QuerySpec([(Q.COUNT, -1)], filter=flt3)).stats[1])
                            out[-1]["entries_match_oracle"] = bool(r3.stats[1] == out[-1]["oracle_num_entries_scanned_in_filter"])
                            out[-1]["oracle_entries_s"] = time.perf_counter() - t_o
                        else:
                            out[-1]["entries_match_oracle"] = None
                # the same query from a caller that does not read the exact statistic (PG_QUERY_STATS_UPPER_BOUND_OK: no transducer, no pass)
                if want(vid + "-bound"):
                    spb = Q.QuerySpec([(Q.SUM, 0)], filter=flt3, stats_upper_bound_ok=True)
                    report(vid + "-bound", "a leap-frogging filter at 1 B rows, the statistic's upper bound accepted (PG_QUERY_STATS_UPPER_BOUND_OK)", sql, n,
                           B(v) + B(f) + B(k) + (B(b) if vid not in ("AND-NOT-scan", "NOT-NOT-scan") else 0), g, seg, spb)
                    out[-1]["filter_entries_exact"] = bool(g.execute(spb).filter_entries_exact)
            if want("COUNT-filter"):
                report("COUNT-filter", "filter only", "SELECT COUNT(*) WHERE f < 100", n, B(f), g, seg, Q.QuerySpec([(Q.COUNT, -1)], filter=flt))
            if out:
                out[-1]["setup"] = setup
        del seg, k, a, b, a_irr, v_irr, v_win

    # ---- C5: inverted-index AND of 3 postings -> docIds -> SUM, sparse (C = 16 / 64 / 256) and dense (C = 2 / 4 / 8) ----
    for vid, cards, seeds, picks in (("C5-sparse", (16, 64, 256), (11, 12, 13), (3, 5, 7)), ("C5-dense", (2, 4, 8), (21, 22, 23), (1, 2, 5))):
        if not (want(vid) or want(vid + "-count") or (vid == "C5-sparse" and (want("C5-scan-count") or want("C5-scan-count-in-list")))):
            continue
        t0 = time.time()
        cols = []
        for name, card, seed in zip("pqr", cards, seeds):
            ids = S.synthetic_dict_ids(seed, 0, n_c5, card)
            cols.append(S.Column.from_dict_ids(name, np.arange(card, dtype=np.int32), ids, with_inverted=True))
            del ids
        if n_c5 == n:
            v5 = v
        else:
            v5 = S.Column.synthetic_uniform("v", n_c5, v_dictionary("affine"), seed=1)
        seg5 = S.SegmentData(vid, n_c5, cols + [v5])
        gen_s = time.time() - t0
        inv = lambda c, d: Q.leaf(Q.Pred.dict_range(c, d, d + 1, inverted=True))
        scan = lambda c, d: Q.leaf(Q.Pred.dict_range(c, d, d + 1))
        post = [int(c.inverted.nbytes / c.cardinality) for c in cols]
        with engine.open(seg5) as g:
            survivors = int(n_c5 / (cards[0] * cards[1] * cards[2]))
            value_bytes = min(B(v5), survivors * 64)
            if want(vid):
                report(vid, "BASELINE.json configs[4]", "SELECT SUM(v) WHERE p=%d AND q=%d AND r=%d via inverted indexes (C = %d / %d / %d)" % (picks + cards), n_c5,
                       sum(post) + value_bytes, g, seg5, Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(inv(0, picks[0]), inv(1, picks[1]), inv(2, picks[2]))),
                       oracle_spec=Q.QuerySpec([(Q.SUM, 3)], filter=Q.and_(scan(0, picks[0]), scan(1, picks[1]), scan(2, picks[2]))),
                       cpu_rows=n_c5,
                       extra={"posting_bytes_read": post, "value_bytes": value_bytes, "host_generate_s": gen_s,
                              "algorithmic_bytes_note": "the three serialized postings + min(B(v), survivors x 64 B); dense intermediates are not charged (BASELINE.md C5)"})
            if want(vid + "-count"):
                report(vid + "-count", "BASELINE.json configs[4], two postings, COUNT", "SELECT COUNT(*) WHERE p=%d AND q=%d via inverted indexes" % picks[:2], n_c5, post[0] + post[1], g, seg5,
                       Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(inv(0, picks[0]), inv(1, picks[1]))),
                       oracle_spec=Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(scan(0, picks[0]), scan(1, picks[1]))))
            if vid == "C5-sparse" and want("C5-scan-count"):
                # the same three predicates by SCANNING the 4 / 6 / 8-bit columns (no inverted index): scan_narrow_kernel, four tiles per wave and iteration
                sspec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(scan(0, picks[0]), scan(1, picks[1]), scan(2, picks[2])))
                report("C5-scan-count", "BASELINE.json configs[4] without the inverted indexes, COUNT", "SELECT COUNT(*) WHERE p=%d AND q=%d AND r=%d by scanning p, q, r (%d / %d / %d bits)"
                       % (picks + tuple(c.bits for c in cols)), n_c5, sum(B(c) for c in cols), g, seg5, sspec)
            if vid == "C5-sparse" and want("C5-scan-count-in-list"):
                # IN lists over the same narrow columns: dictId-set leaves in scan_narrow_kernel (one register up to five bits, eight words of LDS above)
                inl = lambda c, members: Q.leaf(Q.Pred.dict_set(c, members, cols[c].cardinality))
                ispec = Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(inl(0, [1, 3, 6, 12]), inl(1, list(range(0, 64, 5))), inl(2, list(range(1, 256, 9)))))
                report("C5-scan-count-in-list", "BASELINE.json configs[4] without the inverted indexes, IN lists, COUNT", "SELECT COUNT(*) WHERE p IN (4 values) AND q IN (13 values) AND r IN (29 values) by scanning p, q, r (%d / %d / %d bits)"
                       % tuple(c.bits for c in cols), n_c5, sum(B(c) for c in cols), g, seg5, ispec)
        del seg5, cols

    # ---- C1: 10 M rows, raw int32 forward index (BASELINE.json configs[0] is the reference's CPU case; COUNT(*) itself is O(1)) ----
    if any(want(x) for x in ("C1-count-range", "C1-sum", "C1-count", "C1-group-by", "C1-group-by-raw-double")):
        n1 = 10_000_000
        raw = S.Column.raw("raw_i32", S.synthetic_dict_ids(42, 0, n1, 1_000_000))
        seg1 = S.SegmentData("c1", n1, [raw])
        with engine.open(seg1) as g:
            report("C1-count-range", "BASELINE.json configs[0], scan-forcing companion", "SELECT COUNT(*) WHERE raw_i32 BETWEEN 1 AND 10 (10 M rows, raw)", n1, 4 * n1, g, seg1,
                   Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.raw_range(0, 1, 10))), cpu_rows=n1)
            report("C1-sum", "BASELINE.json configs[0], scan-forcing companion", "SELECT SUM(raw_i32) (10 M rows, raw)", n1, 4 * n1, g, seg1, Q.QuerySpec([(Q.SUM, 0)]), cpu_rows=n1)
            report("C1-count", "BASELINE.json configs[0] literally: O(1) in the reference (NonScanBasedAggregationOperator) and here", "SELECT COUNT(*) (10 M rows)", n1, 0, g, seg1,
                   Q.QuerySpec([(Q.COUNT, -1)]))
        if want("C1-group-by"):
            # the C3 query on ONE 10 M-row segment: since round 5 a single launch (the one-item form of the batch's group-by kernel)
            kc = S.Column.synthetic_uniform("k", n1, np.arange(1000, dtype=np.int32) * 3, seed=31)
            ac = S.Column.synthetic_uniform("a", n1, (np.arange(100000, dtype=np.int64) * 5 + 1).astype(np.int32), seed=32)
            bc = S.Column.synthetic_uniform("b", n1, np.arange(65536, dtype=np.int32) * 2, seed=33)
            segg = S.SegmentData("c1g", n1, [kc, ac, bc])
            with engine.open(segg) as g:
                report("C1-group-by", "BASELINE.json configs[2]'s query on one 10 M-row segment", "SELECT SUM(a), MAX(b) GROUP BY k (1000 groups, 10 M rows)", n1, B(kc) + B(ac) + B(bc), g, segg,
                       Q.QuerySpec([(Q.SUM, 1), (Q.MAX, 2)], group_by=[0]))
        if want("C1-group-by-raw-double"):
            # the same query keyed by a raw (no-dictionary) DOUBLE column: NoDictionarySingleColumnGroupKeyGenerator.java:100-135 keys it by value; here the
            # first query builds the column's dictionary + rank image on the device (pg_unit_rank_image.hip), every later one reads the image
            rng = np.random.default_rng(34)
            values = np.sort(rng.normal(0.0, 1e6, 1000))
            dc = S.Column.raw_typed("d", values[S.synthetic_dict_ids(35, 0, n1, 1000)].astype(np.float64))
            ac = S.Column.synthetic_uniform("a", n1, (np.arange(100000, dtype=np.int64) * 5 + 1).astype(np.int32), seed=32)
            bc = S.Column.synthetic_uniform("b", n1, np.arange(65536, dtype=np.int32) * 2, seed=33)
            segd = S.SegmentData("c1gd", n1, [dc, ac, bc])
            with engine.open(segd) as g:
                report("C1-group-by-raw-double", "BASELINE.json configs[2]'s query on one 10 M-row segment, keyed by a raw DOUBLE column", "SELECT SUM(a), MAX(b) GROUP BY d (1000 distinct doubles, 10 M rows)",
                       n1, B(dc) + B(ac) + B(bc), g, segd, Q.QuerySpec([(Q.SUM, 1), (Q.MAX, 2)], group_by=[0]),
                       extra={"algorithmic_bytes_note": "charged with the raw column's 8 bytes per doc (what the reference reads); the kernel reads the 10-bit rank image built by the first query"})
    # ---- the small-segment regime: 64 segments of 10 M rows (BASELINE.json configs[0]'s size), one query over all of them ----
    # (a) pg_execute_batch: one launch, every segment folds its own result; (b) the way BaseCombineOperator would drive pg_execute: 16
    # host threads, each with the next segment, every pg_execute on a stream of its own; (c) one pg_execute after the other.
    c1x64_ids = ("C1x64-count-range", "C1x64-dict-sum", "C1x64-dict-sum-irregular", "C1x64-dict-sum-in-list", "C1x64-group-by", "C5x64", "C5x64-count")
    if any(want(x) for x in c1x64_ids):
        import ctypes as C
        import threading
        n1, nseg = 10_000_000, 64
        t0 = time.time()
        segs = []
        for s in range(nseg):
            raw = S.Column.raw("raw_i32", S.synthetic_dict_ids(4200 + s, 0, n1, 1_000_000))
            fcol = S.Column.synthetic_uniform("f", n1, np.arange(1000, dtype=np.int32), seed=7000 + s)
            vcol = S.Column.synthetic_uniform("v", n1, (np.arange(100000, dtype=np.int64) * 7 + 3 + s).astype(np.int32), seed=8000 + s)
            # (columns 3, 4: the same sum through a dictionary WITHOUT structure -- the normal case of a real Pinot dictionary -- and a 1000-value key)
            wcol = S.Column.synthetic_uniform("w", n1, v_dictionary("irregular", seed=9000 + s), seed=8500 + s)
            kcol = S.Column.synthetic_uniform("k", n1, np.arange(1000, dtype=np.int32), seed=9500 + s)
            cols = [raw, fcol, vcol, wcol, kcol]
            if want("C5x64") or want("C5x64-count"):
                # (columns 5, 6, 7: BASELINE.json configs[4]'s three inverted-index columns, C = 16 / 64 / 256, on every small segment)
                for name, card, seed in (("p", 16, 11000 + s), ("q", 64, 12000 + s), ("r", 256, 13000 + s)):
                    cols.append(S.Column.from_dict_ids(name, np.arange(card, dtype=np.int32), S.synthetic_dict_ids(seed, 0, n1, card), with_inverted=True))
            segs.append(S.SegmentData("c1_%d" % s, n1, cols))
        gen_s = time.time() - t0
        opened = [engine.open(sd) for sd in segs]
        try:
            shapes = [("C1x64-count-range", "SELECT COUNT(*) WHERE raw_i32 BETWEEN 1 AND 10", lambda sd: Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.raw_range(0, 1, 10))), lambda sd: B(sd.columns[0])),
                      ("C1x64-dict-sum", "SELECT SUM(v) WHERE f < 100", lambda sd: Q.QuerySpec([(Q.SUM, 2)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))), lambda sd: B(sd.columns[1]) + B(sd.columns[2])),
                      ("C1x64-dict-sum-irregular", "SELECT SUM(w) WHERE f < 100 (w: 100000 sorted distinct values from the whole int32 range, another set per segment)",
                       lambda sd: Q.QuerySpec([(Q.SUM, 3)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))), lambda sd: B(sd.columns[1]) + B(sd.columns[3])),
                      ("C1x64-group-by", "SELECT SUM(v), MAX(f) GROUP BY k (1000 groups)", lambda sd: Q.QuerySpec([(Q.SUM, 2), (Q.MAX, 1)], group_by=[4]),
                       lambda sd: B(sd.columns[1]) + B(sd.columns[2]) + B(sd.columns[4]))]
            # an IN list of 100 of f's 1000 values (every third of the first 300): a dictId-set leaf -- the words ride in the batch's blob (round 6)
            shapes.append(("C1x64-dict-sum-in-list", "SELECT SUM(v) WHERE f IN (100 values)",
                           lambda sd: Q.QuerySpec([(Q.SUM, 2)], filter=Q.leaf(Q.Pred.dict_set(1, list(range(0, 300, 3)), 1000))), lambda sd: B(sd.columns[1]) + B(sd.columns[2])))
            inv = lambda c, d: Q.leaf(Q.Pred.dict_range(c, d, d + 1, inverted=True))
            posting = lambda sd, c: int(sd.columns[c].inverted.nbytes / sd.columns[c].cardinality)
            shapes += [("C5x64", "SELECT SUM(v) WHERE p=3 AND q=5 AND r=7 via inverted indexes (C = 16 / 64 / 256)",
                        lambda sd: Q.QuerySpec([(Q.SUM, 2)], filter=Q.and_(inv(5, 3), inv(6, 5), inv(7, 7))),
                        lambda sd: posting(sd, 5) + posting(sd, 6) + posting(sd, 7) + int(sd.num_docs / (16 * 64 * 256)) * 64),
                       ("C5x64-count", "SELECT COUNT(*) WHERE p=3 AND q=5 via inverted indexes", lambda sd: Q.QuerySpec([(Q.COUNT, -1)], filter=Q.and_(inv(5, 3), inv(6, 5))),
                        lambda sd: posting(sd, 5) + posting(sd, 6))]
            for vid, sql, mk, nb in shapes:
                if not want(vid):
                    continue
                specs = [mk(sd) for sd in segs]
                nbytes = sum(nb(sd) for sd in segs)
                handles = (C.c_void_p * nseg)(*[g.handle for g in opened])
                queries = (C.POINTER(_abi.pg_query) * nseg)(*[C.pointer(sp.c) for sp in specs])
                results = (_abi.pg_result * nseg)()
                statuses = (C.c_int * nseg)()
                modes = {}

                call_wall = [0.0]

                def run_batch():
                    # the wall clock of a batch is the library call's: 64 ctypes frees and status checks in a Python loop (~1 us each)
                    # are the harness's, not the server's
                    t0 = time.perf_counter()
                    st = engine.execute_batch_raw(handles, queries, nseg, results, statuses)
                    call_wall[0] = (time.perf_counter() - t0) * 1e3
                    assert st == _abi.PG_OK
                    ms = 0.0        # the ONE shared launch, apportioned over the items by the library: their sum is its duration
                    for i in range(nseg):
                        ms += results[i].device_ms
                        assert statuses[i] == _abi.PG_OK
                        engine.lib.pg_result_free(C.byref(results[i]))
                    return ms

                def run_serial():
                    r = _abi.pg_result()
                    for g, sp in zip(opened, specs):
                        assert g.execute_raw(sp, r) == _abi.PG_OK
                        engine.lib.pg_result_free(C.byref(r))
                    return 0.0

                def run_threads(nthreads=16):
                    def work(t):
                        r = _abi.pg_result()
                        for i in range(t, nseg, nthreads):
                            assert opened[i].execute_raw(specs[i], r) == _abi.PG_OK      # ctypes releases the GIL inside the call
                            engine.lib.pg_result_free(C.byref(r))
                    ts = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
                    [t.start() for t in ts]
                    [t.join() for t in ts]
                    return 0.0

                sweep = [("batch_bpc%s" % b, run_batch) for b in os.environ.get("PINOT_BENCH_BATCH_SWEEP", "").split(",") if b]
                for mode, fn in [("batch", run_batch), ("worker_threads", run_batch), ("python_threads16", run_threads), ("serial", run_serial)] + sweep:
                    if mode.startswith("batch_bpc"):
                        engine.reinit(PINOT_GPU_BATCH_BLOCKS_PER_CU=mode[len("batch_bpc"):])
                    if mode == "worker_threads":      # the same call with the shared launch off: every segment a pg_execute of its own on the library's worker threads
                        engine.reinit(PINOT_GPU_BATCH_LAUNCH=0)
                    for _ in range(5):
                        fn()
                    walls, dev = [], []
                    for _ in range(steps):
                        t0 = time.perf_counter()
                        d = fn()
                        walls.append(call_wall[0] if fn is run_batch else (time.perf_counter() - t0) * 1e3)
                        dev.append(d)
                    modes[mode] = {"wall_ms": sum(walls) / len(walls), "wall_ms_min": min(walls), "aggregate_GBps": nbytes / (sum(walls) / len(walls)) / 1e6,
                                   "frac_of_8TBps": nbytes / (sum(walls) / len(walls)) / 1e6 / HBM_PEAK_GBPS}
                    if mode == "worker_threads":
                        engine.reinit(PINOT_GPU_BATCH_LAUNCH=None)
                    if mode.startswith("batch_bpc"):
                        engine.reinit(PINOT_GPU_BATCH_BLOCKS_PER_CU=None)
                    if mode.startswith("batch"):
                        modes[mode]["kernel_ms"] = sum(dev) / len(dev)
                        modes[mode]["kernel_GBps"] = nbytes / (sum(dev) / len(dev)) / 1e6 if sum(dev) > 0 else None
                exact = None
                # (the body that ran, as the library reports it per item; the shared launch is scan_lean_batch_kernel for the simple / raw shapes, scan_private_batch_kernel otherwise)
                body = engine.execute_batch(opened[:1], specs[:1])[0][1].dominant_kernel
                launch = {"scan_simple_kernel": "scan_lean_batch_kernel", "scan_raw_kernel": "scan_lean_batch_kernel", "scan_hist_kernel": "scan_hist_batch_kernel",
                          "group_private_kernel": "group_lds_batch_kernel", "index_and_kernel": "index_and_batch_kernel"}.get(body, "scan_private_batch_kernel")
                if check:
                    got = engine.execute_batch(opened, specs)
                    exact = True
                    for sd, sp, (st, res) in zip(segs, specs, got):
                        wanted = oracle.execute(sd, sp) if vid.startswith("C5x64") else oracle.execute_sliced(sd, sp)
                        if vid.startswith("C5x64"):
                            exact = exact and st == _abi.PG_OK and [(a.count, a.sum_i64) for a in res.aggregations] == [(a.count, a.sum_i64) for a in wanted.aggregations] and res.stats[0] == wanted.stats[0]
                            continue
                        exact = exact and st == _abi.PG_OK and bool(oracle.matches_sliced(res, wanted, [f for f, _ in sp.aggregations]) and res.stats[0] == wanted["docs_scanned"])
                out.append({"id": vid, "config": "BASELINE.json configs[0] x 64 segments: the small-segment regime of a real server", "query": sql + " over 64 segments of 10 M rows",
                            "rows": n1 * nseg, "algorithmic_bytes": int(nbytes), "modes": modes, "kernel": launch, "kernel_body": body, "kernel_ms": modes["batch"].get("kernel_ms"),
                            "all_kernels_ms": modes["batch"].get("kernel_ms"), "step_ms_host_clock": modes["batch"]["wall_ms"], "achieved_GBps": modes["batch"]["aggregate_GBps"],
                            "frac": modes["batch"]["frac_of_8TBps"], "frac_dominant_kernel": (modes["batch"]["kernel_GBps"] / HBM_PEAK_GBPS) if modes["batch"].get("kernel_GBps") else None,
                            "frac_host_clock": modes["batch"]["frac_of_8TBps"],
                            "rows_per_s": n1 * nseg / modes["batch"]["wall_ms"] * 1e3, "bit_exact_vs_oracle": exact, "host_generate_s": gen_s,
                            "note": "frac / achieved_GBps are on the HOST clock around the whole call (lowering, launch, completion of all 64 segments)"})
        finally:
            [g.close() for g in opened]
    return out
