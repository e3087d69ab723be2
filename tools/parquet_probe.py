#!/usr/bin/env python3
"""Times the Parquet scan leaf on one GPU: writes a synthetic file with pyarrow (the headline table:
k Int64 over 10 values, a/b/c Float64, plus a low-cardinality string), then scans it.

usage: parquet_probe.py [rows] [reps] [compression] [dictionary 0|1] [row_group_rows]
Prints wall time of a bare scan (file in the page cache), the device time per kernel, and the
headline plan (filter -> project -> group by) run straight off the file.
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
qe = importlib.import_module("query-engine_amd")


def main():
    import pyarrow as pa
    import pyarrow.parquet as pq
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    compression = sys.argv[3] if len(sys.argv) > 3 else "none"
    dictionary = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
    rg = int(sys.argv[5]) if len(sys.argv) > 5 else 8_000_000
    path = f"/tmp/qe_probe_{n}_{compression}_{int(dictionary)}.parquet"
    rng = np.random.default_rng(42)
    k = rng.integers(0, 10, n)
    t0 = time.perf_counter()
    tbl = pa.table({"k": pa.array(k), "a": pa.array(rng.random(n) * 2 - 1), "b": pa.array(rng.random(n) * 2 - 1), "c": pa.array(rng.random(n)),
                    "s": pa.array(np.char.add("group-", k.astype(str)))})
    pq.write_table(tbl, path, compression=compression, use_dictionary=["k", "s"] if dictionary else False, row_group_size=rg)
    size = os.path.getsize(path)
    print(f"wrote {path}: {size / 1e6:.1f} MB in {time.perf_counter() - t0:.1f} s", flush=True)

    ctx = qe.Context(0)
    C = qe.ColumnExpression

    def timed(label, make, cols_bytes):
        list(make().execute())
        ctx.synchronize()
        ctx.profile(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            out = list(make().execute())
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        prof = ctx.profile_read()
        ctx.profile(False)
        print(f"{label}: {dt * 1e3:.2f} ms  {n / dt / 1e9:.3f} G rows/s  file {size / dt / 1e9:.2f} GB/s  decoded {cols_bytes / dt / 1e9:.2f} GB/s  {prof}",
              flush=True)
        return out

    class Streamed:
        """A consumer that takes a row group at a time and lets go of it."""
        def __init__(self, plan):
            self.plan = plan

        def execute(self):
            for batch in self.plan.execute():
                del batch
            return []

    timed("scan k,a,b,c, batches kept", lambda: qe.ParquetScanExec(ctx, path, ["k", "a", "b", "c"]), n * 32)
    timed("scan k,a,b,c", lambda: Streamed(qe.ParquetScanExec(ctx, path, ["k", "a", "b", "c"])), n * 32)
    timed("scan a", lambda: qe.ParquetScanExec(ctx, path, ["a"]), n * 8)
    timed("scan k", lambda: qe.ParquetScanExec(ctx, path, ["k"]), n * 8)
    timed("scan s", lambda: qe.ParquetScanExec(ctx, path, ["s"]), n * 11)

    def headline():
        scan = qe.ParquetScanExec(ctx, path, ["k", "a", "b", "c"])
        sel = qe.SelectionExec([scan], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
        proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
        return qe.AggregateExec([proj], [C(0)], [qe.SumExpression(C(1)), qe.CountExpression(None)], group_capacity_hint=64)

    out = timed("headline plan off the file", headline, n * 32)
    print("groups", out[0].num_rows)
    os.remove(path)


if __name__ == "__main__":
    main()
