#!/bin/bash
# Per-kernel register / scratch / LDS use of every gfx950 kernel in libpinot_gpu.so, from the compiler's own remarks
# (-Rpass-analysis=kernel-resource-usage); no GPU needed.  A kernel with scratch is a kernel that spills: DESIGN.md §4.3f is what that cost once.
#   tools/register_audit.sh [out.tsv]      (default profiles/kernel_resource_usage.tsv)
set -u
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$(realpath -m "${1:-$root/profiles/kernel_resource_usage.tsv}")"
cd "$root/pinot_amd/csrc"
tmp=$(mktemp -d)
units=$(sed -n 's/^GPU_UNITS := //p' Makefile)
for u in $units; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -Rpass-analysis=kernel-resource-usage -c -o /dev/null $u.hip 2> $tmp/$u.txt ) &
done
wait
python3 - "$tmp" "$out" <<'PY'
import sys, re, glob, os, subprocess
tmp, out = sys.argv[1], sys.argv[2]
rows = []
for f in sorted(glob.glob(tmp + "/*.txt")):
    unit = os.path.basename(f)[:-4]
    cur = None
    for line in open(f):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m: cur = {"unit": unit, "name": m.group(1)}; rows.append(cur); continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)", line)
        if m and cur is not None: cur[m.group(1).strip()] = int(m.group(2))
# kernels defined `static` in a shared header are compiled into every unit that includes it: one row each (the first unit's), when the numbers agree
seen = {}
unique = []
for r in rows:
    key = (r["name"], r.get("VGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS Size"))
    if key in seen:
        seen[key]["units"] += 1
        continue
    r["units"] = 1
    seen[key] = r
    unique.append(r)
rows = unique
# (rocPRIM's sort / select instantiations behind the rank image are the library's kernels, not this engine's: left out, as in tests/test_gpu_kernel_coverage.py)
rows = [r for r in rows if "rocprim" not in r["name"]]
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
with open(out, "w") as o:
    o.write("unit\tunits_with_the_same_code\tkernel\tVGPRs\tAGPRs\tSGPRs\tscratch_bytes_per_lane\tVGPR_spills\tSGPR_spills\toccupancy_waves_per_SIMD\tLDS_bytes\n")
    for r, d in zip(rows, dem):
        d = re.sub(r"^void ", "", d)
        o.write("%s\t%d\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n" % (r["unit"], r["units"], d[:200], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))
spill = [(r, d) for r, d in zip(rows, dem) if r.get("ScratchSize", 0) > 0]
print("%d kernels, %d with scratch, %d with VGPR spills" % (len(rows), len(spill), sum(1 for r, _ in spill if r.get("VGPRs Spill", 0) > 0)))
for r, d in spill: print("  scratch %4d B/lane, %3d VGPR spills, %3d VGPRs:" % (r["ScratchSize"], r.get("VGPRs Spill", 0), r.get("VGPRs", 0)), r["unit"], d[:140])
PY
[ -n "${KEEP_TMP:-}" ] && echo "$tmp" || rm -rf "$tmp"
