#!/usr/bin/env python3
"""Small single-GPU driver for profiling one configuration under rocprofv3.

usage: probe.py <c2|c3|c5|c4|c4b|c5m|c5mp|p|utf8> [rows] [reps] [batches]   (the rows in that many equal batches, default 1; c5m: the c5 plan over 1 M uniform keys; utf8: the reference's
test_filter shape at 100 M rows, bench.py's utf8_filter_config, with the deferred row count)
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

qe = importlib.import_module("query-engine_amd")


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64_000_000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    n_batches = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = qe.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    if cfg == "utf8":
        import json
        import bench
        ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC | qe.FLAG_DEFERRED_ROW_COUNT)
        r = bench.utf8_filter_config(qe, ctx, torch, dev)
        print("utf8 rows", r["rows"], "ms", r["ms_per_step"], "bytes_per_row", r["bytes_per_row"], "frac", r["frac_hbm_peak_end_to_end"], json.dumps(r["kernels_ms_per_step"]))
        return
    k = torch.empty(n, dtype=torch.int64, device=dev)
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)
    c = torch.empty(n, dtype=torch.float64, device=dev)
    if cfg in ("c4", "c4b"):
        ctx.generate(2, 77, 0, n, 1_000_000, 1.1, k.data_ptr())
    elif cfg in ("c5m", "c5mp"):
        ctx.generate(1, 99, 0, n, 0, int(os.environ.get("QE_PROBE_KEYS", "1000000")), k.data_ptr())
    else:
        ctx.generate(3, 43, 0, n, 0, 0, k.data_ptr())
    ctx.generate(0, 44, 0, n, -1.0, 1.0, a.data_ptr())
    ctx.generate(0, 45, 0, n, -1.0, 1.0, b.data_ptr())
    ctx.generate(0, 46, 0, n, 0.0, 1.0, c.data_ptr())
    ctx.synchronize()
    C = qe.ColumnExpression
    per = (n + n_batches - 1) // n_batches
    batches = [qe.Batch.from_torch(ctx, [t[i:i + per] for t in (k, a, b, c)]) for i in range(0, n, per)]

    def plan():
        scan = qe.ScanExec(ctx, batches)
        if cfg == "c2":
            sel = qe.SelectionExec([scan], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
            return qe.ProjectionExec([sel], [qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
        if cfg == "c3":
            return qe.AggregateExec([scan], [C(0)], [qe.SumExpression(C(3)), qe.CountExpression(None)], group_capacity_hint=64)
        if cfg == "c4":
            return qe.AggregateExec([scan], [C(0)], [qe.SumExpression(C(3)), qe.MinExpression(C(3)), qe.MaxExpression(C(3)),
                                                      qe.AvgExpression(C(3))], group_capacity_hint=1_100_000)
        if cfg == "c4b":  # the same keys, accumulators over two value columns: the log carries two value columns, not (key, value) pairs
            return qe.AggregateExec([scan], [C(0)], [qe.SumExpression(C(3)), qe.SumExpression(C(2)), qe.MinExpression(C(3)), qe.MaxExpression(C(2))],
                                    group_capacity_hint=int(os.environ.get("QE_PROBE_HINT", "1100000")))
        if cfg == "p":
            return qe.ProjectionExec([scan], [qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
        sel = qe.SelectionExec([scan], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
        proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
        if cfg == "c5mp":  # a rank's share of the plan over a million keys: a PARTIAL aggregate, hinted as bench.py hints it
            return qe.AggregateExec([proj], [C(0)], [qe.SumExpression(C(1)), qe.CountExpression(None)], mode=qe.AGG_MODE_PARTIAL, group_capacity_hint=int(os.environ.get("QE_PROBE_HINT", "1100000")))
        return qe.AggregateExec([proj], [C(0)], [qe.SumExpression(C(1)), qe.CountExpression(None)],
                                group_capacity_hint=int(os.environ.get("QE_PROBE_HINT", "1100000")) if cfg == "c5m" else 64)

    def run():
        try:
            return list(plan().execute())[0]
        except qe.QeError:
            if not any(v in os.environ for v in ("QE_SPILL_ABLATE_FIRST", "QE_DIRECT_ABLATE", "QE_PART_ABLATE")):  # (a library built for timing experiments may leave nothing to return)
                raise

            class Nothing:
                num_rows = -1
            return Nothing()

    out = run()
    torch.cuda.synchronize()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(cfg, "rows", n, "ms", dt * 1e3, "Grows/s", n / dt / 1e9, "out_rows", out.num_rows, ctx.profile_read())


if __name__ == "__main__":
    main()
