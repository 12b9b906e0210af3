"""CPU test: the C++ host mirror (parser, predicate lowering, writers, segment-directory loader) under AddressSanitizer + UBSan,
through the standalone drivers of tools/asan (no Python in the sanitized process)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import segment_dirs as D
from test_segment_loader_cpu import _synthetic_columns

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with the sanitizer runtimes")
def test_host_mirror_is_clean_under_asan_and_ubsan(tmp_path):
    n, k, cols = _synthetic_columns()
    good = D.write_v3(tmp_path, "seg_v3", n, cols)
    kcol = cols[0]
    ids = np.searchsorted(kcol.dict_values, k).astype(np.int32)
    sorted_v1 = D.write_v1(tmp_path, "seg_v1_sorted", n, cols, sorted_fwd={"k": D.sorted_forward_index(ids, kcol.cardinality)})
    broken = os.path.join(str(tmp_path), "broken")
    shutil.copytree(good, broken)
    with open(os.path.join(broken, "v3", "columns.psf"), "wb") as f:
        f.write(b"\0" * 100)
    truncated = os.path.join(str(tmp_path), "truncated")
    shutil.copytree(sorted_v1, truncated)
    with open(os.path.join(truncated, "v.sv.unsorted.fwd"), "wb") as f:
        f.write(b"\0" * 10)
    env = dict(os.environ, TMPDIR=str(tmp_path))
    out = subprocess.run([os.path.join(ROOT, "tools", "asan", "run.sh"), good, sorted_v1, broken, truncated, os.path.join(str(tmp_path), "missing")],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode("utf-8", "replace")
    assert out.returncode == 0, text[-4000:]
    assert "asan: clean" in text and "AddressSanitizer" not in text and "runtime error" not in text, text[-4000:]
    assert "seg_v3: ok" in text and "broken: status 1" in text and "missing: status 1" in text
