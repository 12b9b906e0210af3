#!/usr/bin/env python3
"""bench.py -- filtered SUM over 1 B-row dictionary-encoded segments, one segment per GPU (BASELINE.json configs[1]/[3]).

Workload C2b: SELECT SUM(v) FROM t WHERE f < 100
  v: INT, dictionary {7k+3 : k < 100000} -> 17-bit fixed-bit forward index (2.125 GB), dictIds uniform, seed 2r+1
  f: INT, dictionary {0..999}            -> 10-bit fixed-bit forward index (1.25 GB),  dictIds uniform, seed 2r+2
  predicate lowered to dictId range [0, 100) (10 % selectivity); r = rank (segment r lives on GPU r).
A step = one pg_execute over the whole resident segment (fused scan -> filter -> SUM kernel reading f's dictIds and
v's device-built value plane (`roofline.kernel` names the kernel that ran) + a one-block partial reduction +
200-byte readback + stream sync).  Columns are generated on the host by the product's C++ writer
in Pinot's on-disk layout and copied to HBM by pg_segment_open before the timed region.

Launch:  python bench.py --gpus 1 --steps 20 --warmup 3
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Prints ONE JSON line on rank 0.  At N=1 the line also carries `cpu_baseline` (the C oracle on one host core over the same full
workload -- it doubles as the parity check) and `cpu_baseline_all_cores` (the workload split into one segment per host core, the way
the reference's combine operator runs segments).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s float4-copy ceiling)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=int(os.environ.get("PINOT_BENCH_ROWS", 1_000_000_000)))
    ap.add_argument("--threshold", type=int, default=100, help="f < threshold (dictIds [0, threshold) of 1000)")
    ap.add_argument("--dictionary", default="affine", choices=["affine", "irregular", "window"],
                    help="dictionary of v: {7k+3} (BASELINE.md C2), 100000 sorted distinct values from the whole int32 range, or from a 2^20 window")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-settle", action="store_true", help="skip the 48 untimed launches that step through the GPU clock transient")
    ap.add_argument("--extra", action="store_true", help="also time the other BASELINE.md query shapes (stderr)")
    ap.add_argument("--profile-waves", action="store_true", help="diagnostic: per-wave phase cycle counters (perturbs timing slightly)")
    return ap.parse_args()


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from pinot_amd import _abi
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine

    n = args.rows
    t0 = time.time()
    v = S.Column.synthetic_uniform("v", n, v_dictionary(args.dictionary), seed=2 * rank + 1)
    f = S.Column.synthetic_uniform("f", n, np.arange(1000, dtype=np.int32), seed=2 * rank + 2)
    seg = S.SegmentData("c2b_%d" % rank, n, [v, f])
    gen_s = time.time() - t0
    spec = Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, args.threshold)))
    algorithmic_bytes = v.fwd.nbytes + f.fwd.nbytes   # B(f) + B(v) = 3.375 B/row

    engine = Engine(device_id=local_rank, time_kernels=True, profile_waves=args.profile_waves)
    t0 = time.time()
    gseg = engine.open(seg)
    h2d_s = time.time() - t0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = _abi.pg_result()
    lib = engine.lib

    def step():
        st = gseg.execute_raw(spec, res)
        if st != _abi.PG_OK:
            raise RuntimeError(lib.pg_last_error().decode())
        out = (res.aggregations[0].sum_i64, res.aggregations[0].count, res.dominant_kernel_ms, res.device_ms, int(res.dominant_kernel))
        if args.profile_waves:
            step.cycles = [int(c) for c in res.profile_cycles] + [int(res.profile_waves)]
        lib.pg_result_free(C.byref(res))
        return out

    # Clock settle: the first ~35 launches after an idle period run through the GPU's power-management transient (0.72 -> 0.58 ->
    # 0.69 -> 0.575 ms for this kernel, tools/steps_probe.py); a resident query engine is never in that state, so it is stepped through
    # before the W warm-up steps.  Reported in the JSON ("clock_settle_launches"); --no-clock-settle turns it off.
    settle = 0 if args.no_clock_settle else 48
    for _ in range(settle):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    device_ms = []
    last = None
    for _ in range(args.steps):
        last = step()
        kernel_ms.append(last[2])
        device_ms.append(last[3])
    barrier()
    elapsed = time.perf_counter() - t0
    from pinot_amd import distributed as D
    elapsed = D.max_over_ranks(elapsed, "cuda")
    # host-side merge of the per-segment partials (no data-path collective: 16 bytes per rank travel)
    merged_sum, merged_count = D.merge_sum_count(D.gather_partials([last[0], last[1]], "cuda"))

    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
    # HBM traffic per launch comes from a separate rocprofv3 --pmc pass (it cannot be collected inside this process);
    # the committed summary applies to the default workload only.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and os.environ.get("PINOT_GPU_VALUE_PLANE", "-1") != "0":
        t = json.load(open(tpath)).get(_abi.KERNEL_NAMES[last[4]], {})
        if t.get("workload_rows") == n and args.threshold == 100:
            traffic = t.get("bytes_per_launch")
    result = None
    if rank == 0:
        rows_per_s = world * n * args.steps / elapsed
        achieved = algorithmic_bytes / (avg_kernel_ms * 1e-3) / 1e9
        result = {
            "metric": "scanned rows/sec + achieved HBM GB/s, filtered SUM on 1B-row segment",
            "value": rows_per_s,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": "C2b: SELECT SUM(v) WHERE f < t, %d rows/segment, v 17-bit dict (C=100000), f 10-bit dict (C=1000), "
                                   "selectivity %.0f%%, one segment per GPU, host-side merge" % (n, args.threshold / 10.0),
                       "rows_per_segment": n, "segments": world, "algorithmic_bytes_per_row": algorithmic_bytes / n},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "kernel": _abi.KERNEL_NAMES[last[4]], "kernel_ms": avg_kernel_ms,
                         "algorithmic_bytes_per_launch": algorithmic_bytes},
            "dictionary": args.dictionary,
            "clock_settle_launches": settle,
            "hbm_GBps_whole_step": world * algorithmic_bytes * args.steps / elapsed / 1e9,
            "wave_profile": ({"waves": step.cycles[4], "cycles_per_wave": {"memory_wait": step.cycles[0] / step.cycles[4], "filter": step.cycles[1] / step.cycles[4],
                                                                          "aggregate": step.cycles[2] / step.cycles[4], "loop_total": step.cycles[3] / step.cycles[4]}}
                             if args.profile_waves else None),
            "result": {"sum": merged_sum, "count": merged_count},
            "setup": {"host_generate_s": gen_s, "segment_open_h2d_s": h2d_s, "device_bytes": gseg.device_bytes(),
                      "h2d_GBps": gseg.device_bytes() / h2d_s / 1e9, "host_threads": S.host_threads()},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            ores = _abi.pg_result()
            t0 = time.perf_counter()
            rc = oracle.execute_raw(seg, spec, ores)
            cpu_s = time.perf_counter() - t0
            assert rc == 0
            osum, ocount = ores.aggregations[0].sum_i64, ores.aggregations[0].count
            oracle.load().po_result_free(C.byref(ores))
            result["cpu_baseline"] = {"value": n / cpu_s, "unit": "rows/s", "cores": 1, "kind": "port",
                                      "sample": "the full workload (%d rows, same segment, same query) through the C oracle on one host core "
                                                "(one segment = one thread, as in BaseCombineOperator); %.1f s" % (n, cpu_s),
                                      "host_cores_available": os.cpu_count()}
            result["parity"] = {"bit_exact_vs_oracle": bool(osum == last[0] and ocount == last[1]), "oracle_sum": osum, "gpu_sum": last[0]}
            result["cpu_baseline_all_cores"] = cpu_baseline_all_cores(seg, spec, n, osum, ocount)
    if args.extra and rank == 0 and world == 1:
        run_extra(gseg, seg, n, lib, Q, _abi, C)
    gseg.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def v_dictionary(kind, cardinality=100000):
    """The dictionary of the summed column: BASELINE.md's arithmetic progression, or sorted distinct values without structure."""
    import numpy as np
    if kind == "affine":
        return (np.arange(cardinality, dtype=np.int64) * 7 + 3).astype(np.int32)
    rng = np.random.default_rng(20260921)
    lo, hi = (-2 ** 31, 2 ** 31 - 1) if kind == "irregular" else (0, 2 ** 20)
    vals = np.unique(rng.integers(lo, hi, 4 * cardinality, dtype=np.int64))
    return np.sort(rng.permutation(vals)[:cardinality]).astype(np.int32)


def cpu_baseline_all_cores(seg, spec, n, want_sum, want_count):
    """the reference runs one segment per thread (BaseCombineOperator), so the honest all-core number splits
    the workload into as many equal segments as the host has cores.  The same packed columns are sliced at multiples of 8 docs
    (8 docs of a b-bit column are b whole bytes), the C oracle runs on every slice in its own thread, the partials are merged."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from pinot_amd import _abi
    from pinot_amd import segment as S
    cores = os.cpu_count() or 1
    bounds = [min(n, ((n * i // cores) + 7) // 8 * 8) for i in range(cores)] + [n]
    slices = []
    for i in range(cores):
        lo, hi = bounds[i], bounds[i + 1]
        if hi <= lo:
            continue
        cols = []
        for c in seg.columns:
            first = lo * c.bits // 8
            cols.append(S.Column(c.name, c.encoding, c.bits, c.cardinality, c.fwd[first:first + ((hi - lo) * c.bits + 7) // 8 + 8], c.dictionary, None, c.dict_values,
                                 stored_type=c.stored_type))
        slices.append(S.SegmentData("slice%d" % i, hi - lo, cols))

    def run(part):
        res = _abi.pg_result()
        rc = oracle.execute_raw(part, spec, res)
        out = (rc, int(res.aggregations[0].sum_i64), int(res.aggregations[0].count)) if rc == 0 else (rc, 0, 0)
        oracle.load().po_result_free(C.byref(res))
        return out
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        parts = list(pool.map(run, slices))
    cpu_s = time.perf_counter() - t0
    ok = all(rc == 0 for rc, _, _ in parts) and sum(p[1] for p in parts) == want_sum and sum(p[2] for p in parts) == want_count
    return {"value": n / cpu_s, "unit": "rows/s", "cores": cores, "kind": "port", "merged_result_matches": bool(ok),
            "sample": "the full workload split into %d equal segments, one oracle thread per segment, partials merged; %.2f s" % (len(slices), cpu_s)}


def run_extra(gseg, seg, n, lib, Q, _abi, C):
    """Other BASELINE.md shapes on the same resident segment (diagnostics on stderr, not the bench line)."""
    shapes = {
        "C2a SUM(v) WHERE v in 10% range": (Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(0, 45000, 55000))), seg.columns[0].fwd.nbytes),
        "C2a 50%": (Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(0, 25000, 75000))), seg.columns[0].fwd.nbytes),
        "C2a 90%": (Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(0, 5000, 95000))), seg.columns[0].fwd.nbytes),
        "C2b 1%": (Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 10))), seg.columns[0].fwd.nbytes + seg.columns[1].fwd.nbytes),
        "C2b 50%": (Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 500))), seg.columns[0].fwd.nbytes + seg.columns[1].fwd.nbytes),
        "COUNT WHERE f<100": (Q.QuerySpec([(Q.COUNT, -1)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))), seg.columns[1].fwd.nbytes),
        "MAX(v) WHERE f<100": (Q.QuerySpec([(Q.MAX, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, 100))), seg.columns[0].fwd.nbytes + seg.columns[1].fwd.nbytes),
        "SUM(v),COUNT GROUP BY f": (Q.QuerySpec([(Q.SUM, 0), (Q.COUNT, -1)], group_by=[1]), seg.columns[0].fwd.nbytes + seg.columns[1].fwd.nbytes),
        "MAX(v) GROUP BY f": (Q.QuerySpec([(Q.MAX, 0)], group_by=[1]), seg.columns[0].fwd.nbytes + seg.columns[1].fwd.nbytes),
    }
    res = _abi.pg_result()
    for name, (spec, nbytes) in shapes.items():
        ms = []
        for i in range(6):
            st = gseg.execute_raw(spec, res)
            if st != _abi.PG_OK:
                print("extra %s failed: %s" % (name, lib.pg_last_error().decode()), file=sys.stderr)
                break
            if i:
                ms.append(res.dominant_kernel_ms)
            lib.pg_result_free(C.byref(res))
        if ms:
            k = sum(ms) / len(ms)
            print(json.dumps({"extra": name, "kernel_ms": k, "rows_per_s": n / k * 1e3, "GBps": nbytes / k / 1e6}), file=sys.stderr)


if __name__ == "__main__":
    main()
