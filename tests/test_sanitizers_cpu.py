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
    # mutated SQL for the sanitized parser / combine table (the same mutations as test_sql_parser_never_crashes_on_mutated_queries)
    rng = np.random.default_rng(7)
    seeds = ["SET enableNullHandling = true; SET minServerGroupTrimSize = 1; SET groupTrimThreshold = 2; SELECT k1, SUM(a), AVG(b) AS v FROM t GROUP BY k1, k2 ORDER BY v DESC NULLS LAST, k2, COUNT(*) LIMIT 2",
             "SELECT COUNT(*), MIN(a) FROM t WHERE a > 1 AND (b IN (1, 2) OR NOT c BETWEEN 5 AND 9) GROUP BY k1, k2 ORDER BY MAX(a), k1 DESC LIMIT 3"]
    tokens = ["SELECT", "FROM", "WHERE", "GROUP", "BY", "ORDER", "LIMIT", "AND", "OR", "NOT", "NULLS", "FIRST", "DESC", "AS", "SET", "(", ")", ",", ";", "*", "=", "'", "1", "-1", "k1", "k2", "SUM", "é"]
    sql_file = os.path.join(str(tmp_path), "mutated.sql")
    with open(sql_file, "w", encoding="utf-8") as f:
        for seed_sql in seeds:
            f.write(seed_sql + "\n")
            words = seed_sql.split(" ")
            for _ in range(300):
                w = list(words)
                for _ in range(int(rng.integers(1, 3))):
                    op, at = int(rng.integers(0, 3)), int(rng.integers(0, len(w)))
                    if op == 0 and len(w) > 1:
                        del w[at]
                    elif op == 1:
                        w.insert(at, tokens[int(rng.integers(0, len(tokens)))])
                    else:
                        w[at] = tokens[int(rng.integers(0, len(tokens)))]
                f.write(" ".join(w) + "\n")
    env = dict(os.environ, TMPDIR=str(tmp_path), PINOT_ASAN_SQL_FILE=sql_file)
    out = subprocess.run([os.path.join(ROOT, "tools", "asan", "run.sh"), good, sorted_v1, broken, truncated, os.path.join(str(tmp_path), "missing")],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode("utf-8", "replace")
    assert out.returncode == 0, text[-4000:]
    assert "asan: clean" in text and "AddressSanitizer" not in text and "runtime error" not in text, text[-4000:]
    assert "sql file: 602 queries" in text
    assert "seg_v3: ok" in text and "broken: status 1" in text and "missing: status 1" in text
