#!/usr/bin/env python3
"""bench.py -- filtered SUM over 1 B-row dictionary-encoded segments (BASELINE.json configs[1] and configs[3]).

Headline workload, C2b of BASELINE.md section 3:   SELECT SUM(v) FROM t WHERE f < 100
  v: INT, dictionary {7k+3 : k < 100000} -> 17-bit fixed-bit forward index (2.125 GB), dictIds uniform, seed 2s+1
  f: INT, dictionary {0..999}            -> 10-bit fixed-bit forward index (1.25 GB),  dictIds uniform, seed 2s+2
  predicate lowered to the dictId range [0, 100) (10 % selectivity); s = segment number.
C4: `--segments S` (default 8) segments IN TOTAL at every N; segment s lives on GPU (s mod N), one process
per GPU, no collective on the data path: the 16-byte partials travel over gloo and are merged on the host (SumAggregationFunction.merge).
A step = one pg_execute per resident segment of the rank, one after the other (fused scan -> filter -> SUM kernel, its records folded
into a pinned 200-byte host record).  `overlapped` reports the same step with the rank's segments in flight together through ONE
pg_execute_batch call (small segments share one launch; segments that fill the chip on their own -- these -- run their own kernels
concurrently on the library's worker threads and streams; PINOT_GPU_BATCH_LAUNCH=0 forces the latter for every size).  value = S * rows * steps / time, "scaling": "strong".  Columns are generated on the host by the
product's C++ writer in Pinot's on-disk layout and copied to HBM by pg_segment_open before the timed region.

Launch:  python bench.py --gpus 1 --steps 20 --warmup 3
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Prints ONE JSON line on rank 0.  At N=1 the line also carries
  cpu_baseline            the C oracle ("port") on one host core over segment 0's full workload -- it doubles as a parity check
  cpu_baseline_all_cores  the same workload split into one segment per host core (how BaseCombineOperator runs segments)
  parity                  every segment's result against the oracle at full size
  variants                every other BASELINE.json configuration, driver-timed with the parity check on: C2b over dictionaries without
                          structure (irregular / 2^20 window), C2a, C3 (config 3) with and without a filter, C5 sparse and dense
                          (config 5), C1 (config 0's scan pair)
  roofline.empirical_peak the box's own 16 B/lane streaming-read ceiling, measured in this run
  roofline.frac           on ALL kernels of the query (HIP events around every launch); frac_dominant_kernel: the scan kernel alone
  cold_launch_ms          one query after >= 1.2 s without a launch, no settle launches (best of three)
  summary                 LAST key: {variant id: [frac on all kernels, all_kernels_ms, bit exact]} for the headline and every variant
The printed line stays under 8 KB (tools/bench_line.py): the `variants` array goes to gpurun_out/bench_variants.json and the uncut
result to gpurun_out/bench_full.json, both named in the line.
Environment: PINOT_BENCH_CHECK_VARIANTS=1 keeps the variants' oracle check on under --no-cpu-baseline; PINOT_BENCH_CHECK_ENTRIES_1B=1 also replays
numEntriesScannedInFilter of the AND-NOT-scan variant on the oracle at 1 B rows (~45 s of one host core); PINOT_BENCH_OUT names the side files' directory.
`--single-process --gpus N`: ONE process drives N devices (segment s on device s mod N, one pg_execute_batch per step) -- the
deployment shape of a Pinot server; the driver's torchrun launch stays one process per GPU.
"""
import argparse
import ctypes as C
import json
import os
import shutil
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
    ap.add_argument("--segments", type=int, default=int(os.environ.get("PINOT_BENCH_SEGMENTS", 8)),
                    help="segments in total (BASELINE.md C4: 8 at every N); segment s runs on GPU s mod N")
    ap.add_argument("--rows", type=int, default=int(os.environ.get("PINOT_BENCH_ROWS", 1_000_000_000)))
    ap.add_argument("--rows-c5", type=int, default=int(os.environ.get("PINOT_BENCH_ROWS_C5", 0)), help="rows of the C5 variants (0 = --rows)")
    ap.add_argument("--threshold", type=int, default=100, help="f < threshold (dictIds [0, threshold) of 1000)")
    ap.add_argument("--dictionary", default="affine", choices=["affine", "irregular", "window"],
                    help="dictionary of v in the headline: {7k+3} (BASELINE.md C2), 100000 sorted distinct values from the whole int32 range, or from a 2^20 window")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--variants", default="", help="regexp: only the variants whose id matches")
    ap.add_argument("--no-clock-settle", action="store_true", help="skip the 48 untimed launches that step through the GPU clock transient")
    ap.add_argument("--single-process", action="store_true",
                    help="the deployment shape: ONE process (a Pinot server is one JVM) drives all --gpus N devices -- segment s resident on device s mod N, "
                         "one pg_execute_batch call per step (launch plain `python bench.py --single-process --gpus N`, no torch.distributed.run)")
    return ap.parse_args()


def v_dictionary(kind, cardinality=100000, seed=20260921):
    """The dictionary of a summed column: BASELINE.md's arithmetic progression, or sorted distinct values without structure."""
    import numpy as np
    if kind == "affine":
        return (np.arange(cardinality, dtype=np.int64) * 7 + 3).astype(np.int32)
    rng = np.random.default_rng(seed)
    lo, hi = (-2 ** 31, 2 ** 31 - 1) if kind == "irregular" else (0, 2 ** 20)
    vals = np.unique(rng.integers(lo, hi, 4 * cardinality, dtype=np.int64))
    return np.sort(rng.permutation(vals)[:cardinality]).astype(np.int32)


def c2b_segment(S, s, n, dictionary):
    v = S.Column.synthetic_uniform("v", n, v_dictionary(dictionary), seed=2 * s + 1)
    f = S.Column.synthetic_uniform("f", n, __import__("numpy").arange(1000, dtype="int32"), seed=2 * s + 2)
    return S.SegmentData("c2b_%d" % s, n, [v, f])


class Timer:
    """Runs a query `steps` times on a resident segment and averages the HIP-event times the engine reports."""

    def __init__(self, lib, _abi):
        self.lib, self._abi = lib, _abi
        self.res = _abi.pg_result()

    def run(self, gseg, spec, steps, warmup):
        kernel, device, wall, kid = [], [], [], -1
        for i in range(warmup + steps):
            t0 = time.perf_counter()
            st = gseg.execute_raw(spec, self.res)
            t1 = time.perf_counter()
            if st != self._abi.PG_OK:
                raise RuntimeError(self.lib.pg_last_error().decode())
            if i >= warmup:
                kernel.append(self.res.dominant_kernel_ms)
                device.append(self.res.device_ms)
                wall.append((t1 - t0) * 1e3)
                kid = int(self.res.dominant_kernel)
            self.lib.pg_result_free(C.byref(self.res))
        mean = lambda x: sum(x) / len(x)
        return {"kernel_ms": mean(kernel), "all_kernels_ms": mean(device), "step_ms_host_clock": mean(wall), "kernel": self._abi.KERNEL_NAMES.get(kid, "")}


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    single = args.single_process
    if single and world != 1:
        raise SystemExit("--single-process is one process: do not launch it through torch.distributed.run")
    if not single and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run (or --single-process)" % (args.gpus, world))
    num_devices = args.gpus if single else 1                 # devices THIS process drives
    physical = max(torch.cuda.device_count(), 1)
    aliased = single and physical < num_devices
    if aliased:
        # fewer HIP devices than --gpus: the N device ids alias the devices there are (id d -> HIP device d mod physical), each id with
        # its own batch contexts, streams and launches -- the multi-device code of pg_execute_batch runs on a one-GPU box (include/pinot_gpu.h pg_device_count)
        os.environ["PINOT_GPU_ALIAS_DEVICES"] = str(num_devices)
    if os.environ.get("PINOT_BENCH_SHARE_GPUS") == "1":
        # test switch: more ranks than GPUs (a one-GPU box running the driver's `torch.distributed.run --nproc-per-node N` line end to end:
        # rendezvous, barrier, MAX over ranks, the gather of the partials, the merge) -- the ranks share the devices there are.  Not a measurement.
        local_rank = local_rank % physical
    torch.cuda.set_device(local_rank)
    if world > 1:
        # no collective on the data path: gloo carries the barrier, the timing MAX and the 16-byte partials (no RCCL needed)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")

    from pinot_amd import _abi
    from pinot_amd import distributed as D
    from pinot_amd import query as Q
    from pinot_amd import segment as S
    from pinot_amd.engine import Engine

    n = args.rows
    num_segments = max(args.segments, world, num_devices)
    mine = [s for s in range(num_segments) if s % world == rank]          # BASELINE.md C4 / SURVEY.md 8(e): segment s -> device s mod N
    t0 = time.time()
    segs = [c2b_segment(S, s, n, args.dictionary) for s in mine]
    if single:
        for s, seg in zip(mine, segs):
            seg.desc.device_id = s % num_devices              # one process, N devices: pg_segment_open places the segment, every call switches to its device
    gen_s = time.time() - t0
    spec = Q.QuerySpec([(Q.SUM, 0)], filter=Q.leaf(Q.Pred.dict_range(1, 0, args.threshold)))
    algorithmic_bytes = segs[0].columns[0].fwd.nbytes + segs[0].columns[1].fwd.nbytes   # B(f) + B(v) = 3.375 B/row

    engine = Engine(device_id=local_rank, time_kernels=True)
    lib = engine.lib
    t0 = time.time()
    gsegs = [engine.open(seg) for seg in segs]
    h2d_s = time.time() - t0
    device_bytes = sum(g.device_bytes() for g in gsegs)

    def barrier():
        for d in range(min(num_devices, physical)):
            torch.cuda.synchronize(d)
        if world > 1:
            dist.barrier()
        for d in range(min(num_devices, physical)):
            torch.cuda.synchronize(d)

    res = _abi.pg_result()
    kernel_ms, query_ms, partials = [], [], [None] * len(gsegs)
    kernel_id = [-1]

    def step(record=False):
        for i, g in enumerate(gsegs):
            st = g.execute_raw(spec, res)
            if st != _abi.PG_OK:
                raise RuntimeError(lib.pg_last_error().decode())
            partials[i] = (int(res.aggregations[0].sum_i64), int(res.aggregations[0].count))
            if record:
                kernel_ms.append(res.dominant_kernel_ms)
                query_ms.append(res.device_ms)               # every kernel of the query (HIP events around all of its launches)
            kernel_id[0] = int(res.dominant_kernel)
            lib.pg_result_free(C.byref(res))

    if single and num_devices > 1:
        # The deployment shape's step: ONE pg_execute_batch over all segments (BaseCombineOperator hands a query's segments to one pool).
        # Items of this size run their own kernel on the library's worker threads, each on its segment's device and stream.
        nall = len(gsegs)
        sp_handles = (C.c_void_p * nall)(*[g.handle for g in gsegs])
        sp_queries = (C.POINTER(_abi.pg_query) * nall)(*[C.pointer(spec.c) for _ in gsegs])
        sp_res = (_abi.pg_result * nall)()
        sp_st = (C.c_int * nall)()

        def step(record=False):      # noqa: F811
            if engine.execute_batch_raw(sp_handles, sp_queries, nall, sp_res, sp_st) != _abi.PG_OK:
                raise RuntimeError(lib.pg_last_error().decode())
            for i in range(nall):
                if sp_st[i] != _abi.PG_OK:
                    raise RuntimeError("batch item %d: status %d" % (i, sp_st[i]))
                partials[i] = (int(sp_res[i].aggregations[0].sum_i64), int(sp_res[i].aggregations[0].count))
                if record:
                    kernel_ms.append(sp_res[i].dominant_kernel_ms)
                    query_ms.append(sp_res[i].device_ms)
                kernel_id[0] = int(sp_res[i].dominant_kernel)
                lib.pg_result_free(C.byref(sp_res[i]))

    # Clock settle: the first ~35 launches after an idle period run through the GPU's power-management transient (0.72 -> 0.58 ->
    # 0.69 -> 0.575 ms for this kernel, tools/steps_probe.py); a resident query engine is never in that state, so it is stepped through
    # before the W warm-up steps.  Reported in the JSON ("clock_settle_launches"); --no-clock-settle turns it off.
    settle = 0 if args.no_clock_settle else 48
    for _ in range((settle + len(gsegs) - 1) // len(gsegs)):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, "cpu")
    # host-side merge of the per-segment partials, in segment order (no data-path collective: 16 bytes per segment travel)
    flat = []
    for i in range((num_segments + world - 1) // world):
        flat += list(partials[i]) if i < len(partials) else [0, 0]
    gathered = D.gather_partials(flat, "cpu")
    per_segment = {}
    for r, vals in enumerate(gathered):
        owned = [s for s in range(num_segments) if s % world == r]
        for i, s in enumerate(owned):
            per_segment[s] = (vals[2 * i], vals[2 * i + 1])
    merged_sum, merged_count = D.merge_sum_count([per_segment[s] for s in range(num_segments)])
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
    avg_query_ms = sum(query_ms) / len(query_ms)
    if single and num_devices > 1:
        # the step's launches overlap (one pg_execute_batch over all devices): an item's own event bracket spans its neighbours' kernels too,
        # so the per-launch figure is the step's wall time apportioned -- launches run back to back on each device
        avg_kernel_ms = avg_query_ms = elapsed / args.steps * 1e3 * min(num_devices, physical) / len(gsegs)
    kernel_name = _abi.KERNEL_NAMES[kernel_id[0]]

    # A query after idle: Pinot's queries arrive whenever they arrive.  >= 1 s without a launch, then ONE query, no settle launches.
    cold = None
    if rank == 0 and world == 1 and not single:
        cold = []
        for _ in range(3):
            time.sleep(1.2)
            t0 = time.perf_counter()
            st = gsegs[0].execute_raw(spec, res)
            wall = (time.perf_counter() - t0) * 1e3
            assert st == _abi.PG_OK
            cold.append({"all_kernels_ms": res.device_ms, "host_clock_ms": wall})
            lib.pg_result_free(C.byref(res))

    # The same step with the rank's segments in flight TOGETHER, the way a server's combine workers would issue them (BaseCombineOperator:
    # one task per segment on a thread pool): (a) one pg_execute_batch call as the library plans it -- items of up to 64 Mi docs share ONE
    # launch, larger ones (the 1 B-row segments of the default run) overlap as launches of their own; (b) the same call with the shared
    # launch switched off.  Reported next to the serial step above (which stays `value`).
    overlapped = None
    if len(gsegs) > 1 and not (single and num_devices > 1):
        nseg = len(gsegs)
        handles = (C.c_void_p * nseg)(*[g.handle for g in gsegs])
        queries = (C.POINTER(_abi.pg_query) * nseg)(*[C.pointer(spec.c) for _ in gsegs])
        bres = (_abi.pg_result * nseg)()
        bst = (C.c_int * nseg)()
        overlapped = {}

        def batch_step():
            if engine.execute_batch_raw(handles, queries, nseg, bres, bst) != _abi.PG_OK:
                raise RuntimeError(lib.pg_last_error().decode())
            ms = bres[0].device_ms
            for i in range(nseg):
                if bst[i] != _abi.PG_OK or (int(bres[i].aggregations[0].sum_i64), int(bres[i].aggregations[0].count)) != partials[i]:
                    raise RuntimeError("batch item %d differs from pg_execute" % i)
                lib.pg_result_free(C.byref(bres[i]))
            return ms

        for mode, env in (("pg_execute_batch", None), ("worker_threads", "0")):
            if env is not None:
                engine.reinit(PINOT_GPU_BATCH_LAUNCH=env)
            for _ in range(max(args.warmup, 2)):
                batch_step()
            barrier()
            t0 = time.perf_counter()
            dev = [batch_step() for _ in range(args.steps)]
            barrier()
            secs = D.max_over_ranks(time.perf_counter() - t0, "cpu")
            overlapped[mode] = {"ms_per_step": secs / args.steps * 1e3, "rows_per_s": num_segments * n * args.steps / secs,
                                "hbm_GBps_whole_step": num_segments * algorithmic_bytes * args.steps / secs / 1e9}
            if env is None:
                overlapped[mode]["kernel_ms"] = sum(dev) / len(dev)
            else:
                engine.reinit(PINOT_GPU_BATCH_LAUNCH=None)

    result = None
    if rank == 0:
        # HBM traffic per launch comes from a separate rocprofv3 --pmc pass (counters cannot be collected inside this process):
        # the committed summary of that pass is replayed here, for the default workload only, and labelled as such.
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.dictionary == "affine" and args.threshold == 100:
            t = json.load(open(tpath)).get(kernel_name, {})
            if t.get("workload_rows") == n:
                traffic = t.get("bytes_per_launch")
                traffic_source = {"replayed": True, "file": "profiles/traffic.json", "from": t.get("source"),
                                  "note": "FETCH_SIZE x2 (gfx950) of a separate rocprofv3 --pmc run of this command; not measured in this run"}
        rows_per_s = num_segments * n * args.steps / elapsed
        achieved = algorithmic_bytes / (avg_query_ms * 1e-3) / 1e9                 # on EVERY kernel of the query (one launch since round 4: the fold is inside the scan)
        achieved_dominant = algorithmic_bytes / (avg_kernel_ms * 1e-3) / 1e9
        result = {
            "metric": "scanned rows/sec + achieved HBM GB/s, filtered SUM on 1B-row segment",
            "value": rows_per_s,
            "unit": "rows/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "process_model": ("one process drives all %d devices (segment s on device s mod N, one pg_execute_batch per step)" % num_devices) if single
                             else "one process per GPU (torch.distributed.run), gloo for the barrier / timing / 16-byte partials",
            "ranks_share_gpus": os.environ.get("PINOT_BENCH_SHARE_GPUS") == "1" and world > physical,
            "aliased_devices": None if not aliased else {"device_ids": num_devices, "hip_devices": physical,
                                                            "note": "PINOT_GPU_ALIAS_DEVICES: the device ids share the physical GPU(s); not a scaling measurement"},
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": "C2b/C4: SELECT SUM(v) WHERE f < t over %d segments x %d rows (segment s on GPU s mod N), v 17-bit dict (C=100000, %s), "
                                   "f 10-bit dict (C=1000), selectivity %.0f%%, host-side merge; one launch = one whole 1 B-row segment"
                                   % (num_segments, n, args.dictionary, args.threshold / 10.0),
                       "rows_per_segment": n, "segments": num_segments, "segments_per_gpu": len(mine), "algorithmic_bytes_per_row": algorithmic_bytes / n,
                       "dictionary": args.dictionary},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_replayed": traffic is not None, "traffic_file": "profiles/traffic.json" if traffic is not None else None,
                         "traffic_source": traffic_source, "kernel": kernel_name, "kernel_ms": avg_kernel_ms,
                         "all_kernels_ms": avg_query_ms, "frac_dominant_kernel": achieved_dominant / HBM_PEAK_GBPS,
                         "frac_note": "achieved / frac are algorithmic bytes over ALL kernels of the query (HIP events around every launch of a pg_execute); "
                                      "frac_dominant_kernel is the scan kernel alone",
                         "launches_timed": len(kernel_ms), "algorithmic_bytes_per_launch": algorithmic_bytes},
            "clock_settle_launches": settle,
            "cold_launch_ms": None if not cold else min(c["all_kernels_ms"] for c in cold),
            "cold_launch": None if not cold else {"samples": cold, "frac": algorithmic_bytes / (min(c["all_kernels_ms"] for c in cold) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                   "note": "one query after >= 1.2 s without a launch, no settle launches, three times; cold_launch_ms = the best of the three"},
            "hbm_GBps_whole_step": num_segments * algorithmic_bytes * args.steps / elapsed / 1e9,
            "overlapped": overlapped,
            "result": {"sum": merged_sum, "count": merged_count},
            "setup": {"host_generate_s": gen_s, "segment_open_h2d_s": h2d_s, "device_bytes": device_bytes,
                      "h2d_GBps": device_bytes / h2d_s / 1e9, "host_threads": S.host_threads()},
        }
        if world == 1:
            peak = C.c_double()
            if lib.pg_measure_stream_read(local_rank, 4 << 30, 6, C.byref(peak)) == _abi.PG_OK:
                result["roofline"]["empirical_peak"] = peak.value
                result["roofline"]["frac_of_empirical_peak"] = achieved / peak.value
                result["roofline"]["empirical_peak_note"] = "best of 6 launches of a pure 16 B/lane read-reduce kernel over 4 GiB, this run, this box"
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            ores = _abi.pg_result()
            t0 = time.perf_counter()
            rc = oracle.execute_raw(segs[0], spec, ores)
            cpu_s = time.perf_counter() - t0
            assert rc == 0
            osum, ocount = int(ores.aggregations[0].sum_i64), int(ores.aggregations[0].count)
            oracle.load().po_result_free(C.byref(ores))
            java = shutil.which("java")
            result["cpu_baseline"] = {"value": n / cpu_s, "unit": "rows/s", "cores": 1, "kind": "port",
                                      "sample": "segment 0's full workload (%d rows, same query) through the C oracle on one host core "
                                                "(one segment = one thread, as in BaseCombineOperator); %.1f s" % (n, cpu_s),
                                      "host_cores_available": os.cpu_count(),
                                      "reference_jvm": {"java_on_this_box": java,
                                                        "note": "no JDK on the measurement box: the reference's JVM path (pinot-perf BenchmarkQueries) cannot be timed here; "
                                                                "kind stays 'port'" if java is None else "a JVM exists here but the reference's jars do not travel with this repository"}}
            ok = [osum == per_segment[0][0] and ocount == per_segment[0][1]]
            t0 = time.perf_counter()
            all_cores = None
            for s in range(num_segments):
                want = oracle.execute_sliced(segs[s], spec)
                got = per_segment[s]
                if s == 0:
                    all_cores = {"value": n / want["seconds"], "unit": "rows/s", "cores": want["threads"], "kind": "port",
                                 "merged_result_matches": bool(want["aggregations"][0]["sum_i64"] == osum and want["aggregations"][0]["count"] == ocount),
                                 "sample": "segment 0's full workload split into %d equal segments, one oracle thread per segment, partials merged; %.2f s"
                                           % (want["slices"], want["seconds"])}
                ok.append(want["aggregations"][0]["sum_i64"] == got[0] and want["aggregations"][0]["count"] == got[1])
            result["cpu_baseline_all_cores"] = all_cores
            result["parity"] = {"bit_exact_vs_oracle": bool(all(ok)), "segments_checked": num_segments, "oracle_sum_segment0": osum, "gpu_sum_segment0": per_segment[0][0],
                                "check_s": time.perf_counter() - t0}
    for g in gsegs[1:]:
        g.close()
    if rank == 0 and world == 1 and not args.no_variants:
        import re
        from tools import bench_variants
        match = re.compile(args.variants) if args.variants else None
        result["variants"] = bench_variants.run(engine, gsegs[0], segs[0], n, args.rows_c5 or n, match,
                                                    check=(not args.no_cpu_baseline) or os.environ.get("PINOT_BENCH_CHECK_VARIANTS") == "1")
        # BASELINE.json configs[0] IS a CPU configuration: its scan-forcing companion's 1-core port figure travels in the line (C3's and C5's: the variants file)
        for v in result["variants"]:
            if v.get("id") == "C1-sum" and v.get("cpu_baseline"):
                result["cpu_baseline_c1"] = dict(v["cpu_baseline"], variant="C1-sum", gpu_rows_per_s_host_clock=v["rows"] / v["step_ms_host_clock"] * 1e3)
    gsegs[0].close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # ONE line under 8 KB (tools/bench_line.py): the variants array and the uncut result go to side files under gpurun_out/, the
        # line keeps the contract's keys + config / roofline / cpu_baseline / parity / overlapped, and `summary` as its LAST key:
        # {configuration: [frac of 8 TB/s on all kernels of the query, all_kernels_ms, bit exact vs oracle]} for BASELINE.json configs[0]..[4]
        from tools import bench_line
        out_dir = os.environ.get("PINOT_BENCH_OUT", os.path.join(ROOT, "gpurun_out"))
        vpath, fpath = bench_line.write_side_files(result, out_dir)
        rel = lambda p: None if p is None else os.path.relpath(p, ROOT)
        print(json.dumps(bench_line.compact(result, rel(vpath) if result.get("variants") else None, rel(fpath))), flush=True)


if __name__ == "__main__":
    main()
