"""bench.py prints ONE line the driver can parse: under 8 KB whatever the number of variants (round 4's 20.9 KB line was not parsed)."""
import json

from tools import bench_line


def synthetic_result(nvariants):
    long_note = "x" * 600
    variants = [{"id": "C%d-some-variant-name" % i, "config": long_note, "query": long_note, "rows": 10 ** 9, "kernel_ms": 0.635817801952362,
                 "all_kernels_ms": 0.635817801952362 + i, "frac": 0.6635155522613199, "bit_exact_vs_oracle": True, "kernel": "scan_hist_kernel",
                 "step_ms_host_clock": 0.7 + i, "frac_host_clock": 0.6012345678,
                 "cpu_baseline": {"value": 2.5e7, "unit": "rows/s", "cores": 1, "kind": "port", "sample": long_note}}
                for i in range(nvariants)]
    return {
        "metric": "scanned rows/sec + achieved HBM GB/s, filtered SUM on 1B-row segment", "value": 1.65e12, "unit": "rows/s", "n_gpus": 1,
        "steps": 20, "warmup": 3, "ms_per_step": 4.8412345678, "higher_is_better": True, "scaling": "strong", "process_model": long_note,
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": long_note, "rows_per_segment": 10 ** 9, "segments": 8, "segments_per_gpu": 8, "algorithmic_bytes_per_row": 3.375, "dictionary": "affine"},
        "roofline": {"bound": "hbm", "achieved": 5769.123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.7211404321, "traffic": 3406000000,
                     "traffic_replayed": True, "traffic_file": "profiles/traffic.json", "traffic_source": {"replayed": True, "note": long_note}, "kernel": "scan_simple_kernel", "kernel_ms": 0.585, "all_kernels_ms": 0.585,
                     "frac_dominant_kernel": 0.72, "frac_note": long_note, "launches_timed": 160, "algorithmic_bytes_per_launch": 3375000000,
                     "empirical_peak": 6520.0, "frac_of_empirical_peak": 0.88, "empirical_peak_note": long_note},
        "clock_settle_launches": 48, "cold_launch_ms": 0.61, "cold_launch": {"samples": [{"all_kernels_ms": 0.6, "host_clock_ms": 0.7}] * 3, "note": long_note},
        "hbm_GBps_whole_step": 5570.0,
        "overlapped": {"pg_execute_batch": {"ms_per_step": 4.7, "rows_per_s": 1.7e12, "hbm_GBps_whole_step": 5700.0, "kernel_ms": 0.58},
                       "worker_threads": {"ms_per_step": 4.8, "rows_per_s": 1.66e12, "hbm_GBps_whole_step": 5600.0}},
        "result": {"sum": 279986541235123, "count": 799942376},
        "setup": {"host_generate_s": 20.1, "segment_open_h2d_s": 3.2, "device_bytes": 27 * 10 ** 9, "h2d_GBps": 8.4, "host_threads": 256},
        "cpu_baseline": {"value": 3.84e8, "unit": "rows/s", "cores": 1, "kind": "port", "sample": long_note, "host_cores_available": 256,
                         "reference_jvm": {"java_on_this_box": None, "note": long_note}},
        "cpu_baseline_c1": {"value": 4.1e8, "unit": "rows/s", "cores": 1, "kind": "port", "sample": long_note, "variant": "C1-sum", "gpu_rows_per_s_host_clock": 3.4e11},
        "cpu_baseline_all_cores": {"value": 1.24e10, "unit": "rows/s", "cores": 256, "kind": "port", "merged_result_matches": True, "sample": long_note},
        "parity": {"bit_exact_vs_oracle": True, "segments_checked": 8, "oracle_sum_segment0": 1, "gpu_sum_segment0": 1, "check_s": 9.0},
        "variants": variants,
    }


def check(line):
    text = json.dumps(line)
    assert len(text) < 8000
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in back
    assert back["config"]["workload"]
    assert back["roofline"]["frac"] and back["roofline"]["bound"] == "hbm" and back["roofline"]["peak"] == 8000.0
    assert back["cpu_baseline"]["kind"] == "port" and back["cpu_baseline"]["cores"] == 1
    assert list(back)[-1] == "summary"
    assert "variants" not in back
    return back


def test_thirty_variants_fit_with_every_requested_object():
    back = check(bench_line.compact(synthetic_result(30), "gpurun_out/bench_variants.json", "gpurun_out/bench_full.json"))
    # [frac on all kernels, all_kernels_ms, bit exact, frac on the host clock]; the headline's fourth entry is the whole step on the host clock
    assert len(back["summary"]) == 31 and back["summary"]["C3-some-variant-name"] == [0.6635, 3.6358, True, 0.6012]
    assert back["summary"]["headline(configs[1],[3])"] == [0.7211, 0.585, True, round(5570.0 / 8000.0, 4)]
    for k in ("cpu_baseline_all_cores", "parity", "cold_launch_ms", "overlapped", "variants_file"):
        assert k in back
    # the traffic figure is a replayed PMC pass and the line says so, with the file it comes from (round 5 stripped the label as a "note")
    assert back["roofline"]["traffic"] == 3406000000 and back["roofline"]["traffic_replayed"] is True and back["roofline"]["traffic_file"] == "profiles/traffic.json"
    # BASELINE.json configs[0] is a CPU configuration: its 1-core port figure is named in the line
    assert back["cpu_baseline_c1"]["kind"] == "port" and back["cpu_baseline_c1"]["cores"] == 1 and back["cpu_baseline_c1"]["variant"] == "C1-sum"
    assert len(back["cpu_baseline_c1"]["sample"]) <= 240


def test_a_very_long_variant_list_is_cut_from_the_best_end_not_the_line():
    back = check(bench_line.compact(synthetic_result(400)))
    assert back["summary"]["_cut"] > 0 and "headline(configs[1],[3])" in back["summary"]


def test_no_variants_and_multi_rank_shape():
    r = synthetic_result(0)
    for k in ("cpu_baseline_all_cores", "parity", "variants", "cold_launch"):
        r.pop(k)
    r["n_gpus"] = 8
    back = check(bench_line.compact(r))
    assert list(back["summary"]) == ["headline(configs[1],[3])"]


def test_side_files(tmp_path):
    r = synthetic_result(3)
    v, f = bench_line.write_side_files(r, str(tmp_path))
    assert len(json.load(open(v))) == 3 and json.load(open(f))["roofline"]["frac_note"]
    assert json.load(open(v))[0]["cpu_baseline"]["kind"] == "port"          # C1 / C3 / C5 carry a 1-core port figure in the variants file
