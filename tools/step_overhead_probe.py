#!/usr/bin/env python3
"""What one rank of an N-GPU run does per step, on one GPU: the PARTIAL plan over 1/N of the rows, the
exchange (the collective replaced by a device copy into every rank's slot, so everything but the
link latency is there) and the FINAL merge.  Prints the step time next to the kernel time.

usage: step_overhead_probe.py [world] [rows_total] [steps] [deferred 0|1]
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

qe = importlib.import_module("query-engine_amd")
D = importlib.import_module("query-engine_amd.distributed")


class CopyDist:
    def __init__(self, world):
        self.world = world

    def all_gather_into_tensor(self, out, inp):
        out.view(self.world, -1).copy_(inp.unsqueeze(0).expand(self.world, -1))


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    total = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    n = total // world
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)  # engine and torch on one stream: nothing else orders them in deferred mode
    ctx = qe.Context(0, stream=work_stream.cuda_stream)
    k = torch.empty(n, dtype=torch.int64, device=dev)
    a, b, c = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
    ctx.generate(3, 43, 0, n, 0, 0, k.data_ptr())
    ctx.generate(0, 44, 0, n, -1.0, 1.0, a.data_ptr())
    ctx.generate(0, 45, 0, n, -1.0, 1.0, b.data_ptr())
    ctx.generate(0, 46, 0, n, 0.0, 1.0, c.data_ptr())
    ctx.synchronize()
    batches = [qe.Batch.from_torch(ctx, [k, a, b, c])]
    C = qe.ColumnExpression
    aggs = lambda col: [qe.SumExpression(C(col)), qe.CountExpression(None)]
    deferred = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
    if deferred:
        ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_DEFERRED_GROUP_COUNT)
    xch = D.PartialExchange(torch, CopyDist(world), dev, world, 1025 if deferred else 64, [torch.int64, torch.float64, torch.int64])
    sys.path.insert(0, ROOT)
    from bench import batch_as_block, column_as_tensor
    bx = {}
    # PROBE_SIDE_STREAM=1: exchange and FINAL merge on a second stream (a second context), so that the next step's
    # PARTIAL kernel overlaps them
    two = os.environ.get("PROBE_SIDE_STREAM") == "1" and deferred
    side = torch.cuda.Stream(device=dev) if two else work_stream
    ctx2 = qe.Context(0, stream=side.cuda_stream) if two else ctx
    if two:
        ctx2.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_DEFERRED_GROUP_COUNT)
    from collections import deque
    inflight = deque()

    marks = {}

    def step():
        t0 = time.perf_counter()
        scan = qe.ScanExec(ctx, batches)
        sel = qe.SelectionExec([scan], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
        proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
        part = list(qe.AggregateExec([proj], [C(0)], aggs(1), mode=qe.AGG_MODE_PARTIAL, group_capacity_hint=64).execute())[0]
        t1 = time.perf_counter()
        if two:
            ready = torch.cuda.Event()
            ready.record(work_stream)
            side.wait_event(ready)
        with torch.cuda.stream(side):
            blk = batch_as_block(torch, part, 3) if deferred else None
            if blk:
                if "x" not in bx:
                    bx["x"] = D.BlockExchange(torch, CopyDist(world), dev, world, blk[0].shape[0], blk[1], blk[2], [torch.int64, torch.float64, torch.int64])
                merged = bx["x"].exchange(blk[0])
            else:
                cols = [column_as_tensor(torch, part.column_desc(i), qe) for i in range(3)]
                merged = xch.exchange_merged(cols)
            t2 = time.perf_counter()
            scan = qe.ScanExec(ctx2, [qe.Batch.from_torch(ctx2, merged)])
            real = qe.SelectionExec([scan], qe.GtExpression(C(2), qe.LiteralInt64Expression(0)))
            out = list(qe.final_plan(real, 1, aggs(1), group_capacity_hint=64).execute())[0]
            if two:
                done = torch.cuda.Event()
                done.record(side)
                inflight.append((part, done))
                if len(inflight) > 2:  # the partial table's block goes back to the first context's pool: not before the side stream has read it
                    _, old = inflight.popleft()
                    work_stream.wait_event(old)
        t3 = time.perf_counter()
        for name, dt in (("partial_plan", t1 - t0), ("exchange_host_side", t2 - t1), ("final_plan", t3 - t2)):
            marks[name] = marks.get(name, 0.0) + dt
        return out

    for _ in range(3):
        out = step()
    torch.cuda.synchronize()
    ctx.synchronize()
    marks.clear()
    ctx.profile(int(os.environ.get("PROBE_PROFILE", "2")))
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = ctx.profile_read()
    kern = sum(v["ms"] for kk, v in prof.items() if kk.startswith("qe_aggregate_static")) / steps
    ctx.synchronize()
    ctx2.synchronize()
    cols = out.columns()
    real = cols[2].values > 0
    assert int(real.sum()) == 10 and int(cols[2].values[real].sum()) * 1 > 0, "merged result is wrong"
    print("count over all ranks' copies", int(cols[2].values[real].sum()))
    print(f"world {world} rows/rank {n} deferred={deferred}: step {dt * 1e3:.3f} ms, headline kernel {kern:.3f} ms, rows out {out.num_rows}")
    print({kk: round(v / steps * 1e3, 3) for kk, v in marks.items()}, "ms per step (host wall, includes waiting for the GPU)")
    print({kk: round(v["ms"] / steps, 4) for kk, v in prof.items()})


if __name__ == "__main__":
    main()
