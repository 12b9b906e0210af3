"""tools/bench_show.py <bench json>: the variants table of one bench.py line."""
import json
import sys

d = json.load(open(sys.argv[1]))
print("value %.4e rows/s  ms_per_step %.4f  kernel_ms %.4f frac %.4f  whole-step GB/s %.0f  parity %s" % (
    d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["hbm_GBps_whole_step"], d.get("parity", {}).get("bit_exact_vs_oracle")))
for mode, o in (d.get("overlapped") or {}).items():
    print("   overlapped %-18s ms_per_step %.4f kernel_ms %s GB/s %.0f" % (mode, o["ms_per_step"], o.get("kernel_ms"), o["hbm_GBps_whole_step"]))
for v in d.get("variants", []):
    print("   %-18s %-26s kernel %.4f all %.4f wall %.4f frac %.3f fdom %s exact=%s %s" % (
        v["id"], v["kernel"], v["kernel_ms"], v["all_kernels_ms"], v["step_ms_host_clock"], v["frac"], v["frac_dominant_kernel"] and round(v["frac_dominant_kernel"], 3),
        v["bit_exact_vs_oracle"], json.dumps(v["modes"]) if v.get("modes") else ""))
