"""Regenerates tests/golden/userdata.npz and userdata_expected.json from the data file the
reference's own tests read (query-engine/src/tests/userdata.parquet, tests/mod.rs:30).

Only the DATA FILE is read (with pyarrow); no reference code is imported or run.  The expected
values that come from the reference's assertions (tests/mod.rs:34-188) are written down in
tests/test_oracle_golden.py with their line numbers; the per-country / per-gender statistics
stored here are an independent pyarrow/numpy cross-check, not reference output.

Run in the build container: python tests/golden/make_userdata_golden.py
"""
import json
import os

import numpy as np
import pyarrow.parquet as pq

SRC = "/root/reference/query-engine/src/tests/userdata.parquet"
HERE = os.path.dirname(os.path.abspath(__file__))

t = pq.read_table(SRC)
out = {}
schema = []
for name in t.column_names:
    col = t.column(name).combine_chunks()
    typ = str(col.type)
    valid = np.array(col.is_valid().to_numpy(zero_copy_only=False), dtype=np.bool_)
    if typ == "string":
        vals = [b"" if v is None else v.encode() for v in col.to_pylist()]
        offs = np.zeros(len(vals) + 1, np.int32)
        np.cumsum([len(v) for v in vals], out=offs[1:])
        out[name + ".offsets"] = offs
        out[name + ".bytes"] = np.frombuffer(b"".join(vals), dtype=np.uint8)
        schema.append([name, "Utf8"])
    elif typ == "int32":
        out[name + ".values"] = np.array([0 if v is None else v for v in col.to_pylist()], dtype=np.int32)
        schema.append([name, "Int32"])
    elif typ == "double":
        out[name + ".values"] = np.array([0.0 if v is None else v for v in col.to_pylist()], dtype=np.float64)
        schema.append([name, "Float64"])
    else:  # timestamp[ns] kept as raw int64 nanoseconds
        out[name + ".values"] = np.array(col.cast("int64").fill_null(0).to_numpy(zero_copy_only=False), dtype=np.int64)
        schema.append([name, "Timestamp(Nanosecond, None)"])
    out[name + ".valid"] = valid
np.savez_compressed(os.path.join(HERE, "userdata.npz"), **out)

# independent cross-check statistics
gender = t.column("gender").to_pylist()
country = t.column("country").to_pylist()
salary = t.column("salary").to_pylist()
ids = t.column("id").to_pylist()
female = [i for i, g in enumerate(gender) if g == "Female"]
stats = {}
for c, s in zip(country, salary):
    d = stats.setdefault(c, {"rows": 0, "nn": 0, "sum": 0.0, "min": None, "max": None})
    d["rows"] += 1
    if s is not None:
        d["nn"] += 1
        d["sum"] += s
        d["min"] = s if d["min"] is None else min(d["min"], s)
        d["max"] = s if d["max"] is None else max(d["max"], s)
expected = {
    "schema": schema,
    "rows": t.num_rows,
    "row_groups": pq.ParquetFile(SRC).num_row_groups,
    "female_row_indices": female,
    "id_sum": int(sum(ids)),
    "salary_nulls": int(sum(1 for s in salary if s is None)),
    "country_stats": stats,
}
with open(os.path.join(HERE, "userdata_expected.json"), "w") as f:
    json.dump(expected, f, indent=1, sort_keys=True)
print("rows", t.num_rows, "female", len(female), "countries", len(stats))
