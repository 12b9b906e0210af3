"""The kernel-coverage gate: every kernel libpinot_gpu.so contains is DISPATCHED by tools/kernel_coverage.py's table, with more tiles than
resident waves, and every entry of the table equals the oracle.

Two regimes, each one `rocprofv3 --kernel-trace` run of the table in a process of its own:
    tiny    100 003 docs, every grid sized for one compute unit (PINOT_GPU_TEST_CUS=1)
    large   12 300 017 docs on the full grid
What must be there is not a hand-kept list: it is the set of device stubs in the built library (`nm -C`), i.e. every `__global__`
instantiation that was compiled in.  A kernel (or a template instance) added without a table entry that reaches it fails here -- in both
regimes, unless EXEMPT below names it with a reason.  (a kernel that was wrong from a wave's second tile on passed the
whole suite for two rounds because no test ran it with two tiles per wave.)
"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

import pytest

from pinot_amd import _abi

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernels the table need not reach in a regime, each with the reason
EXEMPT = {
    "tiny": {},
    "large": {
        # machines of every (states, inputs) class are walked in the tiny regime over ~50 tiles per workgroup; the kernels' tile loop
        # does not depend on the grid (a workgroup's tiles are independent, the chain kernels join them), so the large regime runs a
        # sample of the machines, not all twenty-six classes
        "fsm_tiles_kernel": "sampled in the large regime (all classes in the tiny one)",
        "fsm_tiles_perm_kernel": "sampled in the large regime (all classes in the tiny one)",
        "fsm_tiles_perm8_kernel": "sampled in the large regime (all classes in the tiny one)",
        "build_nullkey_fwd_kernel": "built once per nullable key column from a 300 007-doc segment in both regimes",
    },
}


def kernel_name(text):
    """`void pg::scan_hist_kernel<8, false>(pg::ScanParams) [clone .kd]` -> `scan_hist_kernel<8, false>`."""
    t = text.strip()
    t = re.sub(r"\s*\[clone[^\]]*\]", "", t)
    if t.startswith("void "):
        t = t[5:]
    depth = 0
    for i, ch in enumerate(t):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            t = t[:i]
            break
    return t[4:] if t.startswith("pg::") else t


def kernels_in_the_library():
    out = subprocess.check_output(["nm", "-C", _abi.GPU_LIB_PATH]).decode()
    names = set()
    for line in out.splitlines():
        if "pg::__device_stub__" not in line:
            continue                                    # (rocPRIM's sort / select kernels behind the rank image are the library's, not ours)
        names.add(kernel_name(line.split("__device_stub__", 1)[1]))
    return names


def family(name):
    return name.split("<", 1)[0]


def test_the_name_normalisation():
    assert kernel_name("void pg::scan_hist_kernel<8, false>(pg::ScanParams) [clone .kd]") == "scan_hist_kernel<8, false>"
    assert kernel_name("pg::index_and_kernel(pg::IndexAndParams)") == "index_and_kernel"
    assert kernel_name("void pg::group_partition_aggregate_kernel<0, true>(pg::PartitionParams)") == "group_partition_aggregate_kernel<0, true>"
    lib = kernels_in_the_library()
    assert len(lib) >= 100 and "scan_simple_kernel" in lib and "group_lds_batch_kernel<false>" in lib


@pytest.mark.parametrize("regime", ["tiny", "large"])
def test_every_kernel_is_dispatched_with_more_tiles_than_waves_and_equals_the_oracle(regime, tmp_path):
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    assert os.path.exists(rocprof), "rocprofv3 is part of the image"
    out_dir = str(tmp_path / "trace")
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in list(env):
        if k.startswith("PINOT_GPU_") and k not in ("PINOT_GPU_LIB",):
            del env[k]                                  # the table sets what it needs
    cmd = [rocprof, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "coverage", "--",
           sys.executable, os.path.join(ROOT, "tools", "kernel_coverage.py"), "--regime", regime]
    proc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.startswith("{")]
    assert lines, "no report from the table:\n" + proc.stderr.decode()[-3000:]
    report = json.loads(lines[-1])
    assert report["failed"] == [] and proc.returncode == 0, report["failed"][:10]
    traces = glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True)
    assert traces, "rocprofv3 left no kernel trace under %s" % out_dir
    seen = {}
    for path in traces:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = kernel_name(row["Kernel_Name"])
                seen[name] = seen.get(name, 0) + 1
    ours = {k: v for k, v in seen.items() if k in kernels_in_the_library()}
    missing = sorted(k for k in kernels_in_the_library() if k not in ours and family(k) not in EXEMPT[regime])
    keep = os.environ.get("PINOT_COVERAGE_OUT")
    if keep:
        os.makedirs(keep, exist_ok=True)
        with open(os.path.join(keep, "kernel_coverage_%s.json" % regime), "w") as f:
            json.dump({"report": report, "dispatches": dict(sorted(ours.items())), "missing": missing,
                       "exempt": EXEMPT[regime]}, f, indent=1)
    assert missing == [], "kernels of libpinot_gpu.so the %s table never dispatched: %s" % (regime, missing)
    for fam in EXEMPT[regime]:
        assert any(family(k) == fam for k in kernels_in_the_library()), "EXEMPT names a kernel that no longer exists: " + fam
        if fam.startswith("fsm_"):
            assert any(family(k) == fam for k in ours), "not even a sample of %s ran" % fam


# The environment switches of libpinot_gpu.so that change HOW a query is launched without changing which kernels exist: the
# gate above runs the defaults, and the entries of its table set the switches that are needed to REACH a kernel.  These two runs put the whole
# table through the other side of everything else, in two combinations, and hold every answer against the oracle -- a switch a deployment
# can set is code a deployment can run.  (No dispatch-set requirement here: with the launch structure changed, other kernels answer.)
FLIPPED = {
    "launch-structure": "PINOT_GPU_GROUP_ONE_LAUNCH=0,PINOT_GPU_INDEX_GATHER=0,PINOT_GPU_FSM_EPISODES=0,PINOT_GPU_LEAN_BATCH=0,PINOT_GPU_BATCH_HIST=0,PINOT_GPU_BATCH_GROUP=0,"
                        "PINOT_GPU_BATCH_MORE=0,PINOT_GPU_BATCH_INDEX=0,PINOT_GPU_DIRECT_RESULT=0,PINOT_GPU_INDEX_AND_WAVES=-1,PINOT_GPU_PLAN_CACHE=0,PINOT_GPU_FOLD_ONE_COUNTER=0,PINOT_GPU_LEAP2=0,"
                        "PINOT_GPU_LANE_SKIP=0,PINOT_GPU_SCAN_SPARSE=0,PINOT_GPU_SCAN_RAW=0,PINOT_GPU_SCAN_NARROW=0,PINOT_GPU_RAW64_COALESCED=0,PINOT_GPU_GROUP_PACK=0,"
                        "PINOT_GPU_GROUP_REPLICAS=0,PINOT_GPU_PARTITION_STATS_CACHE=0,PINOT_GPU_PLANE_GCD=0,PINOT_GPU_STAGED_H2D=0,PINOT_GPU_SMALL_BLOCKS_PER_CU=0,"
                        "PINOT_GPU_WIDE_BLOCKS=1,PINOT_GPU_GROUP_WAVES=4",
    "geometry": "PINOT_GPU_GROUP_PUBLISH=0,PINOT_GPU_SET_LDS=0,PINOT_GPU_POLL_RESULT=0,PINOT_GPU_SCAN_NARROW_SINGLE=0,PINOT_GPU_INDEX_AND_WAVES=3,PINOT_GPU_SPARSE_LANES=64,PINOT_GPU_BLOCKS_PER_CU=2,PINOT_GPU_BATCH_BLOCKS_PER_CU=2,"
                "PINOT_GPU_DOUBLE_BUFFER=1,PINOT_GPU_TILE_STEPS=16,PINOT_GPU_GROUP_REPLICAS=1,PINOT_GPU_GROUP_WAVES=8,PINOT_GPU_FOLD_FINALIZE=0,PINOT_GPU_VALUE_PLANE=1,"
                "PINOT_GPU_PLANE_BUDGET_BYTES=1000000",
}


@pytest.mark.parametrize("name", sorted(FLIPPED))
def test_the_table_with_the_other_side_of_the_switches_equals_the_oracle(name):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in list(env):
        if k.startswith("PINOT_GPU_") and k not in ("PINOT_GPU_LIB",):
            del env[k]
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_coverage.py"), "--regime", "tiny", "--fsm-trees", "120", "--flip", FLIPPED[name]],
                          cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.startswith("{")]
    assert lines, "no report from the table:\n" + proc.stderr.decode()[-3000:]
    report = json.loads(lines[-1])
    assert report["failed"] == [] and proc.returncode == 0, report["failed"][:10]
    assert report["entries"] > 60


def test_every_switch_of_the_library_has_a_test_on_its_other_side():
    """Every PINOT_GPU_* name the library reads is set by some test or by the coverage table (or is a trace / test-harness switch): a switch
    added without one fails here."""
    import glob as _glob
    import re as _re
    names = set()
    for path in _glob.glob(os.path.join(ROOT, "pinot_amd", "csrc", "*.h*")) + _glob.glob(os.path.join(ROOT, "pinot_amd", "csrc", "*.hip")):
        names.update(_re.findall(r'"(PINOT_GPU_[A-Z0-9_]+)"', open(path).read()))
    assert len(names) > 40
    tested = ""
    for path in _glob.glob(os.path.join(ROOT, "tests", "*.py")) + [os.path.join(ROOT, "tools", "kernel_coverage.py"), os.path.join(ROOT, "pinot_amd", "engine.py"), os.path.join(ROOT, "pinot_amd", "_abi.py")]:
        tested += open(path).read()
    diagnostics = {"PINOT_GPU_BATCH_TRACE", "PINOT_GPU_EXEC_TRACE", "PINOT_GPU_FSM_TRACE", "PINOT_GPU_PARTITION_TRACE", "PINOT_GPU_RANK_TRACE"}      # stderr only
    missing = sorted(n for n in names if n not in tested and n not in diagnostics)
    assert missing == [], missing
