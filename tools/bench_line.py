"""The ONE stdout line of bench.py, kept under a fixed size.

Round 4's line was 20.9 KB (a 22-entry `variants` array) and the driver's parser saw only its 8 KB tail, so its record carried no
roofline and no cpu_baseline.  The full result now goes to a side file and the printed line keeps the contract's keys, `config`,
`roofline` (with `traffic_replayed` / `traffic_file`: the traffic figure is a committed PMC pass replayed, never this run's), `cpu_baseline`,
`cpu_baseline_all_cores`, `cpu_baseline_c1` (BASELINE.json configs[0] is a CPU configuration: its 1-core port figure), `parity`,
`cold_launch_ms`, `overlapped` and `summary` ({id: [frac on all kernels, all_kernels_ms, bit exact, frac on the host clock]}); free-text notes are
cut, floats are rounded to six significant digits, and keys are dropped in a fixed order if the line would still pass the limit.
No torch / GPU imports here: tests/test_bench_line_cpu.py builds a synthetic result and checks the size.
"""
import json
import os

LIMIT_BYTES = 7600            # the driver keeps an 8 KB tail; stay well inside it

# keys of the printed line, in print order (the contract's keys first; `summary` stays the LAST key)
HEAD_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline", "cpu_baseline", "cpu_baseline_all_cores", "cpu_baseline_c1", "parity", "cold_launch_ms",
             "hbm_GBps_whole_step", "clock_settle_launches", "overlapped", "aliased_devices", "ranks_share_gpus", "result", "setup", "process_model",
             "variants_file", "full_result_file"]
# dropped first -> last when the line is still too long (the contract's keys and roofline / cpu_baseline are never dropped)
DROP_ORDER = ["setup", "process_model", "result", "clock_settle_launches", "ranks_share_gpus", "aliased_devices", "overlapped", "hbm_GBps_whole_step",
              "cpu_baseline_all_cores", "cold_launch_ms", "cpu_baseline_c1", "parity"]
# free text that explains a number: kept in the side file, cut from the line
NOTE_KEYS = {"note", "frac_note", "empirical_peak_note", "traffic_source", "reference_jvm", "host_cores_available", "check_s",
             "oracle_sum_segment0", "gpu_sum_segment0", "launches_timed"}


def _round(x):
    if isinstance(x, float):
        return float("%.6g" % x)
    if isinstance(x, dict):
        return {k: _round(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v) for v in x]
    return x


def _strip_notes(x):
    if isinstance(x, dict):
        return {k: _strip_notes(v) for k, v in x.items() if k not in NOTE_KEYS}
    if isinstance(x, list):
        return [_strip_notes(v) for v in x]
    return x


def _clip(s, n):
    return s if len(s) <= n else s[:n - 3] + "..."


def summary_of(result):
    """{variant id: [frac of 8 TB/s on all kernels of the query, all_kernels_ms, bit exact vs oracle, frac of 8 TB/s on the HOST clock around
    the call]} for the headline and every variant (the headline's fourth entry: the whole step -- all its segments -- on the host clock)."""
    roof = result.get("roofline") or {}
    whole = result.get("hbm_GBps_whole_step")
    out = {"headline(configs[1],[3])": [roof.get("frac"), roof.get("all_kernels_ms"), (result.get("parity") or {}).get("bit_exact_vs_oracle"),
                                        None if whole is None or not roof.get("peak") else whole / roof["peak"]]}
    for v in result.get("variants") or []:
        out[str(v.get("id"))] = [v.get("frac"), v.get("all_kernels_ms"), v.get("bit_exact_vs_oracle"), v.get("frac_host_clock")]
    return {k: [None if a is None else round(a, 4), None if b is None else round(b, 4), c, None if d is None else round(d, 4)] for k, (a, b, c, d) in out.items()}


def compact(result, variants_file=None, full_file=None, limit=LIMIT_BYTES):
    """The dict to print: `result` without the variants array, notes cut, under `limit` bytes once serialised."""
    line = {}
    for k in HEAD_KEYS:
        if k in result:
            line[k] = result[k]
    if variants_file:
        line["variants_file"] = variants_file
    if full_file:
        line["full_result_file"] = full_file
    exact = line.get("result")                  # the merged SUM / COUNT stay as computed
    line = _round(_strip_notes(line))
    if exact is not None:
        line["result"] = exact
    cfg = line.get("config")
    if isinstance(cfg, dict) and isinstance(cfg.get("workload"), str):
        cfg["workload"] = _clip(cfg["workload"], 400)
    for key in ("cpu_baseline", "cpu_baseline_all_cores", "cpu_baseline_c1"):
        if isinstance(line.get(key), dict) and isinstance(line[key].get("sample"), str):
            line[key]["sample"] = _clip(line[key]["sample"], 240)
    summary = summary_of(result)
    line["summary"] = summary

    def size():
        return len(json.dumps(line))

    for k in DROP_ORDER:
        if size() <= limit:
            break
        line.pop(k, None)
    if size() > limit:
        # a very long variants list: keep the headline and the entries furthest below the roofline, say how many were cut
        keep = {"headline(configs[1],[3])": summary["headline(configs[1],[3])"]}
        rest = sorted((k for k in summary if k not in keep), key=lambda k: (summary[k][0] is None, summary[k][0] or 0.0))
        line["summary"] = keep
        for k in rest:
            keep[k] = summary[k]
            if size() > limit:
                del keep[k]
                break
        keep["_cut"] = len(summary) - len(keep)
        line["summary"] = keep
    assert size() <= limit or not line.get("summary"), "bench line over the limit"
    return line


def write_side_files(result, out_dir):
    """variants -> <out_dir>/bench_variants.json, everything -> <out_dir>/bench_full.json; returns the two names (or None, None if unwritable)."""
    try:
        os.makedirs(out_dir, exist_ok=True)
        vpath = os.path.join(out_dir, "bench_variants.json")
        fpath = os.path.join(out_dir, "bench_full.json")
        with open(vpath, "w") as f:
            json.dump(result.get("variants") or [], f, indent=1)
        with open(fpath, "w") as f:
            json.dump(result, f, indent=1)
        return vpath, fpath
    except OSError:
        return None, None
