#!/usr/bin/env python3
"""Writes tests/golden/datatable_v4_golden.json: the DataTable V4 bytes of the reference's golden filtered aggregation
(InnerSegmentAggregationSingleValueQueriesTest.java:44-61 values) as tests/datatable_v4.py -- a Python restatement of DataTableImplV4 /
DataTableBuilderV4 -- encodes them.  No JVM exists here, so these bytes were NOT produced by the reference itself (parity unpinned):
the fixture pins the C++ writer to that restatement and guards both against drifting."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datatable_v4 as D   # noqa: E402
import helpers as H        # noqa: E402

g = H.load_golden_queries()["inner_segment"]["filtered"]
names = ["count(*)", "sum(column1)", "max(column3)", "min(column6)", "avg(column7)"]
types = [D.LONG, D.DOUBLE, D.DOUBLE, D.DOUBLE, D.OBJECT]
row = [g["count"], float(g["sum_column1"]), float(g["max_column3"]), float(g["min_column6"]), (float(g["avg_column7"][0]), g["avg_column7"][1])]
data = D.encode(names, types, [row], D.results_metadata(g["stats"], 1, 1))
out = {"_source": "tools/make_datatable_fixture.py (Python restatement of DataTableImplV4.toBytes; not produced by the reference: parity unpinned)",
       "inner_segment_filtered_aggregation": {"query": "SELECT COUNT(*), SUM(column1), MAX(column3), MIN(column6), AVG(column7) FROM testTable WHERE <BaseSingleValueQueriesTest.FILTER>",
                                              "num_bytes": len(data), "hex": data.hex()}}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "datatable_v4_golden.json"), "w"), indent=1)
print(len(data), "bytes")
