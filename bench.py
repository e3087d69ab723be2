#!/usr/bin/env python3
"""Headline benchmark: filter -> project -> GROUP BY SUM over device-resident synthetic batches.

One "step" = one full execution of the physical plan
    Scan(memory) -> Selection(c > 0.5) -> Projection(k, a + b*c) -> Aggregate(GROUP BY k; SUM(v), COUNT(*))
over this rank's row range of a 1 B-row table (columns k:Int64, a,b,c:Float64 = 32 B/row), through the
C ABI of libqe_gpu.so.  With N ranks the table is split into contiguous row ranges (strong scaling:
the total stays 1 B rows), every rank aggregates its shard in PARTIAL mode, the partial tables are
exchanged with one RCCL all-gather and merged by a FINAL aggregate on every rank.

Prints ONE JSON line (rank 0).  `value` = rows of the whole table / max-over-ranks step time.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
BYTES_PER_ROW = {"c5": 32, "c2": 28, "c3": 16, "c4": 16}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--rows", type=int, default=1_000_000_000, help="rows of the whole table (all ranks)")
    p.add_argument("--groups", type=int, default=10, help="10 = passenger_count-style skewed key; else uniform key domain")
    p.add_argument("--batch-rows", type=int, default=1 << 28)
    p.add_argument("--no-extra", action="store_true", help="skip the secondary single-GPU configurations")
    p.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    p.add_argument("--cpu-rows", type=int, default=48_000_000)
    return p.parse_args()


class DevArray:
    """Raw device memory as a torch tensor (zero copy) through __cuda_array_interface__."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def column_as_tensor(torch, desc, qe):
    ts = {qe.TYPE_INT64: "<i8", qe.TYPE_FLOAT64: "<f8", qe.TYPE_INT32: "<i4"}[desc.type]
    return torch.as_tensor(DevArray(desc.values, desc.length, ts), device="cuda")


def batch_as_block(torch, batch, n_cols):
    """An aggregate's output in deferred mode is one device block with the columns as slices: the block as an
    int64 tensor and each column's offset in 8-byte words, or None when the columns do not lie that way."""
    descs = [batch.column_desc(i) for i in range(n_cols)]
    base = descs[0].values
    offs = [d.values - base for d in descs]
    rows = descs[0].length
    if any(d.validity or d.length != rows for d in descs) or offs != sorted(offs) or any(o % 8 for o in offs):
        return None
    if any(b - a < rows * 8 or b - a > rows * 8 + 1024 for a, b in zip(offs, offs[1:])):
        return None
    words = offs[-1] // 8 + rows
    return torch.as_tensor(DevArray(base, words, "<i8"), device="cuda"), [o // 8 for o in offs], rows


class StdoutToStderr:
    """Whatever native libraries write to file descriptor 1 meanwhile goes to stderr (RCCL prints a version banner on stdout
    when its first communicator comes up): stdout carries the one JSON line and nothing else."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


SAMPLER_CODE = r"""
import glob, json, os, sys, time
out = open(sys.argv[1], "w", buffering=1)
cards = []
for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
    d = os.path.join(c, "device")
    try:
        if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
            continue
    except OSError:
        continue
    hw = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*")))
    cards.append((os.path.basename(os.path.realpath(d)), d, hw[0] if hw else None))
def starred(path):
    try:
        for line in open(path):
            if line.rstrip().endswith("*"):
                return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
    except (OSError, ValueError, IndexError):
        pass
    return None
def number(path):
    try:
        return int(open(path).read().strip())
    except (OSError, ValueError):
        return None
period = float(sys.argv[2])
while True:
    t = time.time()
    rec = {"t": t, "cards": {}}
    for pci, d, hw in cards:
        rec["cards"][pci] = {"sclk_mhz": starred(os.path.join(d, "pp_dpm_sclk")), "mclk_mhz": starred(os.path.join(d, "pp_dpm_mclk")),
                             "power_uw": number(os.path.join(hw, "power1_input")) if hw else None,
                             "temp_mc": number(os.path.join(hw, "temp2_input")) if hw else None,
                             "busy_pct": number(os.path.join(d, "gpu_busy_percent"))}
    out.write(json.dumps(rec) + "\n")
    time.sleep(max(0.0, period - (time.time() - t)))
"""


class ClockSampler:
    """Clocks, power and temperature of the GPUs read off sysfs two hundred times a second by a child process that is started
    before this process has anything to do with a GPU (it touches no HIP: plain file reads of pp_dpm_sclk, pp_dpm_mclk and
    the hwmon files), so that the samples that fall INSIDE the timed steps can be told from those before and after."""

    def __init__(self, period=0.005):
        import subprocess
        import tempfile
        self.path = os.path.join(tempfile.mkdtemp(prefix="qe_bench_clocks_"), "samples.jsonl")
        try:
            self.proc = subprocess.Popen([sys.executable, "-c", SAMPLER_CODE, self.path, str(period)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except OSError:
            self.proc = None

    def window(self, t0, t1, pci_bus_id=None):
        """Summary of the samples taken between the wall-clock times t0 and t1 (time.time()), for the card on PCI bus
        `pci_bus_id` (as torch reports it) or, when that is unknown, every card."""
        out = {"source": "sysfs pp_dpm_sclk / pp_dpm_mclk / hwmon power1_input, temp2_input, sampled by a child process", "window_s": t1 - t0}
        try:
            recs = [json.loads(line) for line in open(self.path) if line.strip()]
        except (OSError, ValueError) as e:
            out["unavailable"] = str(e)[:200]
            return out
        inside = [r for r in recs if t0 <= r["t"] <= t1]
        out["samples_total"], out["samples_in_window"] = len(recs), len(inside)
        cards = {}
        for r in inside:
            for pci, v in r["cards"].items():
                if pci_bus_id is not None:
                    try:
                        if int(pci.split(":")[1], 16) != int(pci_bus_id):
                            continue
                    except (ValueError, IndexError):
                        pass
                cards.setdefault(pci, []).append(v)
        for pci, vs in cards.items():
            summ = {}
            for key, scale, name in (("sclk_mhz", 1, "sclk_mhz"), ("mclk_mhz", 1, "mclk_mhz"), ("power_uw", 1e-6, "power_w"), ("temp_mc", 1e-3, "temp_c"),
                                     ("busy_pct", 1, "busy_pct")):
                xs = [v[key] * scale for v in vs if v.get(key) is not None]
                if xs:
                    summ[name] = {"min": min(xs), "mean": sum(xs) / len(xs), "max": max(xs)}
            out[pci] = summ
        return out

    def pause(self, on):
        """Stops (or lets go on) the sampling child: a leg that is bound by the host's CPUs - the Parquet scan's reader
        threads take every CPU of the process but one - should not share them with a process that polls sysfs."""
        import signal
        if self.proc is not None and self.proc.poll() is None:
            try:
                os.kill(self.proc.pid, signal.SIGSTOP if on else signal.SIGCONT)
            except OSError:
                pass

    def stop(self):
        self.pause(False)
        if self.proc is not None and self.proc.poll() is None:
            self.proc.terminate()
            try:
                self.proc.wait(timeout=5)
            except Exception:
                self.proc.kill()


def load_launcher():
    """query-engine_amd/launcher.py by path: importing the package would load the HIP library, and the ranks must be
    started before this process has anything to do with a GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("qe_launcher", os.path.join(ROOT, "query-engine_amd", "launcher.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    args = parse()
    launcher = load_launcher()
    # QE_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share
    # devices, the exchange goes through host memory); the measured configuration is nccl = RCCL.
    backend = os.environ.get("QE_BENCH_BACKEND", "nccl")
    try:
        world, rank, local_rank, launched = launcher.check_world(args.gpus)
    except launcher.LaunchError as e:
        print(f"bench.py: {e}", file=sys.stderr)
        sys.exit(2)
    sampler = None
    if rank == 0 and (launched or world == 1):
        sampler = ClockSampler()  # a child process, started before this process touches a GPU
    import torch
    n_devices = torch.cuda.device_count()  # does not initialise the device
    if backend == "nccl" and n_devices < world:
        print(f"bench.py: --gpus {world} needs {world} GPUs, this machine has {n_devices}: refusing to run fewer ranks than asked for "
              "(QE_BENCH_BACKEND=gloo rehearses the multi-rank path with ranks sharing a device)", file=sys.stderr)
        sys.exit(2)
    if world > 1 and not launched:
        # nobody started the ranks for us: N fresh children, before this process touches a GPU; rank 0 prints the line
        os.environ["QE_BENCH_SPAWNED"] = "1"
        # a time limit and every rank's stderr in a file: a rendezvous that never completes ends here with a message,
        # the ranks' last words and a non-zero code (QE_BENCH_TIMEOUT seconds, default 480)
        import tempfile
        limit = float(os.environ.get("QE_BENCH_TIMEOUT", "480"))
        logs = os.environ.get("QE_BENCH_LOG_DIR") or tempfile.mkdtemp(prefix="qe_bench_ranks_")
        sys.exit(launcher.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], world, timeout=limit, stderr_dir=logs))
    import torch.distributed as dist

    if backend == "gloo":
        local_rank = local_rank % max(1, n_devices)
    # QE_BENCH_FORCE_DIST=1 runs the multi-rank code path (PARTIAL aggregate, exchange, FINAL merge) with whatever
    # world size the launcher gave, 1 included: a one-GPU box can then exercise the real RCCL calls.
    multi = world > 1 or os.environ.get("QE_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        import datetime
        # (a rendezvous or a collective that never completes raises after this long instead of sitting there until somebody
        # kills the job: QE_BENCH_TIMEOUT seconds, default 480)
        limit = datetime.timedelta(seconds=float(os.environ.get("QE_BENCH_TIMEOUT", "480")))
        with StdoutToStderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
            else:
                dist.init_process_group("gloo", timeout=limit)
            dist.barrier()  # brings torch's communicator up here, inside the redirection
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)
    xdev = dev if backend == "nccl" else torch.device("cpu")
    qe = importlib.import_module("query-engine_amd")
    # One stream for everything: the engine's kernels, torch's copies and the collectives' hand-over.  (torch's
    # default stream is the null stream, whose handle is 0 - "make your own" to the engine - and the two would
    # then be ordered only by the engine's own synchronisations, of which the deferred mode below has none.)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    ctx = qe.Context(local_rank, stream=work_stream.cuda_stream)

    # ---------------------------------------------------------------- data
    total = args.rows
    lo, hi = rank * total // world, (rank + 1) * total // world  # = distributed.shard_range
    n = hi - lo
    k = torch.empty(n, dtype=torch.int64, device=dev)
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)
    c = torch.empty(n, dtype=torch.float64, device=dev)
    seed = 42
    if args.groups == 10:
        ctx.generate(3, seed + 1, lo, n, 0, 0, k.data_ptr())
    else:
        ctx.generate(1, seed + 1, lo, n, 0, args.groups, k.data_ptr())
    ctx.generate(0, seed + 2, lo, n, -1.0, 1.0, a.data_ptr())
    ctx.generate(0, seed + 3, lo, n, -1.0, 1.0, b.data_ptr())
    ctx.generate(0, seed + 4, lo, n, 0.0, 1.0, c.data_ptr())
    ctx.synchronize()

    def batches_of(cols, rows):
        out = []
        for s in range(0, rows, args.batch_rows):
            e = min(rows, s + args.batch_rows)
            out.append(qe.Batch.from_torch(ctx, [t[s:e] for t in cols]))
        return out

    batches = batches_of([k, a, b, c], n)
    C = qe.ColumnExpression
    mode = qe.AGG_MODE_PARTIAL if multi else qe.AGG_MODE_SINGLE
    aggs = lambda col: [qe.SumExpression(C(col)), qe.CountExpression(None)]
    # (the expected number of groups: twice the few-groups case's handful; a tenth above a large count, as the
    # single-GPU configuration over a million keys hints it - the many-groups mode sizes its partitions by the hint,
    # and 512 partitions, what its fastest passes take, hold 1.3 M groups of this shape)
    hint = max(64, 2 * args.groups) if args.groups <= 65536 else args.groups + args.groups // 10

    def local_plan():
        scan = qe.ScanExec(ctx, batches)
        sel = qe.SelectionExec([scan], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
        proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
        return qe.AggregateExec([proj], [C(0)], aggs(1), mode=mode, group_capacity_hint=hint)

    # partial tables (key:int64, sum:float64, count:int64).  Few groups: every rank gets every table
    # (fixed-size all-gather) and merges them all.  Many groups: rows travel to the rank that owns their
    # key (all-to-all), each rank merges its share, the result is the concatenation over the ranks.
    partitioned = args.groups > 65536
    # Few groups: the aggregates run with QE_FLAG_DEFERRED_GROUP_COUNT - their output has one row per table slot (rows
    # past the group count are zero, COUNT(*) = 0) and nothing in a step waits for the device, so the host queues
    # step after step.  The timed region still ends with a barrier and a device synchronise.
    deferred = not partitioned
    default_flags = qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP
    if deferred:
        ctx.set_flags(default_flags | qe.FLAG_DEFERRED_GROUP_COUNT)
    comm, comm_note = None, None
    if multi:
        from importlib import import_module
        D = import_module("query-engine_amd.distributed")
        layout = [torch.int64, torch.float64, torch.int64]
        # The exchange itself runs inside the library (qe_comm_init / qe_exchange_*: its own RCCL communicator, device-side
        # scatter, grouped collectives, no host synchronisation); torch.distributed only carries the rendezvous id, the
        # barrier and the max over ranks.  QE_BENCH_EXCHANGE=torch keeps the exchange in torch.distributed (and is what a
        # gloo rehearsal uses); a library communicator that cannot be set up is reported and the torch path takes over.
        if backend == "nccl" and os.environ.get("QE_BENCH_EXCHANGE", "abi") == "abi":
            try:
                with StdoutToStderr():
                    comm = D.make_comm(qe, ctx, dist, rank, world)
            except qe.QeError as e:
                comm_note = f"qe_comm_init failed ({e}); exchange through torch.distributed"
            ok = torch.tensor([1 if comm else 0], dtype=torch.int64, device=xdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if not int(ok.item()) and comm:
                comm.close()
                comm, comm_note = None, "qe_comm_init failed on another rank; exchange through torch.distributed"
        if partitioned and not comm:
            xch = D.KeyPartitionedExchange(torch, dist, xdev, world, layout)
        elif not comm:
            first = list(local_plan().execute())[0]
            max_g = first.num_rows  # rows of a partial table = slots of its group table
            xch = D.PartialExchange(torch, dist, xdev, world, max_g, layout)
            as_block = batch_as_block(torch, first, 3)
            bxch = D.BlockExchange(torch, dist, xdev, world, as_block[0].shape[0], as_block[1], as_block[2], layout) if as_block else None
            torch.cuda.synchronize()
            del first, as_block
    # QE_BENCH_SIDE_STREAM=1 (an experiment, off by default): exchange and FINAL merge of a step on a second stream (a
    # second engine context), so that the PARTIAL aggregate of the next step - independent work - does not wait behind
    # the collective's latency and the small kernels of the merge.  With one rank over RCCL it measured 3 % slower than
    # one stream (event and stream hand-over costs, nothing to hide); whether a real 8-GPU all-gather's latency
    # turns that around can only be measured on such a node.
    two_streams = multi and deferred and backend == "nccl" and not comm and os.environ.get("QE_BENCH_SIDE_STREAM", "0") == "1"
    if two_streams:
        from collections import deque
        side_stream = torch.cuda.Stream(device=dev)
        ctx2 = qe.Context(local_rank, stream=side_stream.cuda_stream)
        ctx2.set_flags(default_flags | qe.FLAG_DEFERRED_GROUP_COUNT)
        in_flight = deque()
    else:
        side_stream, ctx2 = work_stream, ctx

    def step():
        part = list(local_plan().execute())[0]
        if not multi:
            return part
        return merge(part)

    def merge(part):
        """Exchange of this rank's partial table and the FINAL aggregate over what arrives."""
        if comm:
            # rows nobody filled are zero: COUNT(*) > 0 under the FINAL aggregate keeps the real ones
            if partitioned:
                got = comm.exchange_by_key(part, 1, args.groups)  # a rank's table has at most one row per key of the domain
                cap = max(64, 2 * args.groups // world)
            else:
                got = comm.all_gather(part, part.num_rows)  # deferred mode: one row per table slot, the same number on every rank
                cap = hint
            real = qe.SelectionExec([qe.ScanExec(ctx, [got])], qe.GtExpression(C(2), qe.LiteralInt64Expression(0)))
            return list(qe.final_plan(real, 1, aggs(1), group_capacity_hint=cap).execute())[0]
        cols = [column_as_tensor(torch, part.column_desc(i), qe) for i in range(3)]
        # the only collective of the plan: the exchange of the partial tables over xGMI
        if partitioned:
            table = xch.exchange(cols)
            scan = qe.ScanExec(ctx, [qe.Batch.from_torch(ctx, [c.to(dev) for c in table])] if table[0].shape[0] > 0 else [])
            final = qe.final_plan(scan, 1, aggs(1), group_capacity_hint=max(64, 2 * args.groups // world))
        else:
            # every rank's table in one batch, rows no rank filled are zero: COUNT(*) > 0 keeps the real ones, and the
            # FINAL aggregate merges all ranks in one launch with no trip to the host in between
            if two_streams:
                ready = torch.cuda.Event()
                ready.record(work_stream)
                side_stream.wait_event(ready)  # the partial table is complete before the side stream reads it
            with torch.cuda.stream(side_stream):
                blk = batch_as_block(torch, part, 3) if bxch else None
                merged = bxch.exchange(blk[0]) if blk else xch.exchange_merged(cols)  # the block travels as it is: nothing to pack
                scan = qe.ScanExec(ctx2, [qe.Batch.from_torch(ctx2, [c.to(dev) for c in merged])])
                real = qe.SelectionExec([scan], qe.GtExpression(C(2), qe.LiteralInt64Expression(0)))
                out = list(qe.final_plan(real, 1, aggs(1), group_capacity_hint=hint).execute())[0]
                if two_streams:
                    done = torch.cuda.Event()
                    done.record(side_stream)
                    in_flight.append((part, done))
                    if len(in_flight) > 2:
                        # the partial table's memory goes back to the first context's pool when `part` is dropped:
                        # the work stream must not reuse it before the side stream has read it
                        _, old = in_flight.popleft()
                        work_stream.wait_event(old)
            return out
        return list(final.execute())[0]

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()  # also reports a deferred aggregate that overflowed its table
        if ctx2 is not ctx:
            ctx2.synchronize()

    wall_warm = time.time()
    for _ in range(args.warmup):
        result = step()
    ctx.profile(2)  # HIP events around the main kernel of every large launch only: event records on the small helper launches cost more than those launches
    barrier()
    ctx.profile_read()
    wall0 = time.time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    barrier()
    dt = time.perf_counter() - t0
    wall1 = time.time()
    state_after = gpu_state(local_rank) if rank == 0 else None
    prof = ctx.profile_read()
    ctx.profile(False)
    # the merge alone (exchange + FINAL aggregate of one partial table), timed the same way: what N > 1 adds to a step
    merge_ms = None
    if multi and not two_streams:
        part = list(local_plan().execute())[0]
        merge(part)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            merge(part)
        barrier()
        merge_ms = (time.perf_counter() - t0) / args.steps * 1e3
        del part
    ctx.set_flags(default_flags)
    if ctx2 is not ctx:
        ctx2.set_flags(default_flags)
    topology = {"world_size": world, "backend": "single process"}
    if multi:
        t = torch.tensor([dt, merge_ms or 0.0], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        merge_ms = float(t[1].item()) if merge_ms is not None else None
        here = [None] * world
        dist.all_gather_object(here, {"rank": rank, "device": local_rank, "pid": os.getpid(),
                                      "name": torch.cuda.get_device_name(local_rank),
                                      "pci_bus_id": getattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id", None)})
        topology = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                    "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None,
                    "library_rccl_version": comm.rccl_version if comm else None,
                    "launched_by": "bench.py (child processes)" if os.environ.get("QE_BENCH_SPAWNED") else "external launcher" if launched else "none",
                    "ranks": here}
    if world != args.gpus:
        raise SystemExit(f"bench.py: ran {world} ranks but --gpus {args.gpus} was asked for")
    ms_per_step = dt / args.steps * 1e3
    value = total / (dt / args.steps)

    # dominant kernel: the fused filter+project+aggregate kernel, timed with HIP events on its stream.
    # (The FINAL merge of the N>1 path runs a different, tiny aggregate kernel: it has its own profile entry.)
    kname, kern = dominant_kernel(prof, "qe_aggregate")
    launches = kern["launches"]
    avg_ms = kern["ms"] / launches if launches else float("nan")
    rows_per_launch = n * args.steps / launches if launches else float("nan")  # (a launch may take several batches)
    achieved = BYTES_PER_ROW["c5"] * rows_per_launch / (avg_ms * 1e-3) / 1e9 if launches else float("nan")

    traffic, traffic_src = measured_traffic(kname, rows_per_launch)

    out = {
        "metric": "rows/s + % HBM peak: filter->project->GROUP BY SUM, 1/2/4/8 MI355X",
        "value": value,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "C5: Scan->Selection(c>0.5)->Projection(k, a+b*c)->Aggregate(GROUP BY k; SUM, COUNT)",
                   "rows_total": total, "rows_per_gpu": n, "groups": args.groups, "batch_rows": args.batch_rows,
                   "bytes_per_row": BYTES_PER_ROW["c5"], "partitioning": f"row-range x{world}",
                   "host_sync": "none inside a step (deferred group count); barrier + device synchronise around the timed steps" if deferred
                   else "per aggregate",
                   "streams": "PARTIAL aggregate on one stream, exchange + FINAL merge on a second" if two_streams else "one",
                   "merge": ((("rccl inside the library (qe_exchange_by_key: owner hash + scatter kernel, grouped send / recv)" if partitioned else
                               "rccl inside the library (qe_exchange_all_gather)") if comm else
                              (f"{'rccl' if backend == 'nccl' else backend} through torch.distributed, " +
                               ("all_to_all by key owner" if partitioned else "all_gather"))) + " + FINAL aggregate") if multi else "none",
                   "merge_note": comm_note},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": kname,
                     "avg_launch_ms": avg_ms, "rows_per_launch": rows_per_launch},
        "pct_hbm_peak_per_gpu": 100.0 * (BYTES_PER_ROW["c5"] * n / (dt / args.steps)) / 1e9 / HBM_PEAK_GBS,
        "kernels": prof,
        "merge_ms_per_step": merge_ms,
    }
    out["config"]["topology"] = topology
    out["gpu_state_after_timed_steps"] = state_after
    if sampler is not None:
        # (the window is widened to the warm-up steps in front of the timed ones, which run the same kernels back to back)
        out["gpu_state_during_timed_steps"] = sampler.window(wall_warm, wall1, getattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id", None))
        out["gpu_state_during_timed_steps"]["note"] = "window = warm-up steps + timed steps (%.0f ms of the %.0f ms are the timed steps)" % ((wall1 - wall0) * 1e3, (wall1 - wall_warm) * 1e3)

    cols = result.columns()
    real = np.asarray(cols[2].values) > 0  # deferred mode pads the result to the table's slots
    for col in cols:
        col.values = np.asarray(col.values)[real]
    groups_here = torch.tensor([int(real.sum()), 0], dtype=torch.int64, device=xdev)
    groups_here[1] = int(cols[2].values.sum())
    if multi and partitioned:
        dist.all_reduce(groups_here)  # every rank holds a share of the groups
    if rank == 0:
        order = np.argsort(cols[0].values)
        out["result_groups"] = int(groups_here[0])
        out["result_count_total"] = int(groups_here[1])
        out["result_head"] = [[int(cols[0].values[i]), float(cols[1].values[i]), int(cols[2].values[i])] for i in order[:3]]

    if world == 1 and not args.no_extra:
        global SAMPLER, SAMPLER_PCI
        SAMPLER, SAMPLER_PCI = sampler, getattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id", None)
        out["other_configs"] = extra_configs(qe, ctx, torch, dev, k, a, b, c, n, args)
    if world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.cpu_rows, seed)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if sampler is not None:
        sampler.stop()
    if comm:
        comm.close()
    if multi:
        dist.destroy_process_group()


def gpu_state(device=0):
    """Clocks and power of the device as rocm-smi reports them right now (a child process; nothing here touches HIP):
    recorded next to every timing, since the same kernel measures a few per cent apart from one session to the next."""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
        keep = {}
        for key, val in card.items():
            low = key.lower()
            if any(w in low for w in ("sclk", "mclk", "fclk", "socclk", "power", "junction")):
                keep[key] = val
        return keep
    except Exception as e:  # rocm-smi missing or not answering: the numbers stand without it
        return {"unavailable": str(e)[:200]}


def measured_traffic(kernel_name, rows_per_launch):
    """HBM bytes per launch of the dominant kernel from the PMC passes kept under profiles/ (rocprofv3
    --pmc FETCH_SIZE and --pmc WRITE_SIZE, one run each, FETCH_SIZE doubled as gfx950 needs for 16-byte-
    per-lane streams), scaled from that run's launch size to this one's.  None when no summary matches."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_c5_hbm_traffic.json"))):
        try:
            with open(path) as f:
                summ = json.load(f)["summary"]
        except (OSError, ValueError, KeyError):
            continue
        spec = kernel_name.split("<")[-1].rstrip(">")
        if spec and spec in summ["kernel"]:
            best = (summ["hbm_bytes_per_launch"] / summ["rows_per_launch"] * rows_per_launch, os.path.relpath(path, ROOT))
    return best if best else (None, None)


def dominant_kernel(prof, prefix):
    """The profile entry with the most device time among the kernels whose name starts with `prefix`
    (general interpreter kernel `<prefix>_kernel` or a shape-specialised `<prefix>_static<...>`)."""
    best = (prefix + "_kernel", {"launches": 0, "ms": 0.0})
    for name, rec in prof.items():
        if name.startswith(prefix) and rec["ms"] > best[1]["ms"]:
            best = (name, rec)
    return best


SAMPLER, SAMPLER_PCI, LAST_WINDOW = None, None, None


def timed(ctx, torch, fn, steps, warmup):
    """Seconds per step over `steps` steps with nothing timed inside them, then the same steps once more with HIP events
    around every launch for the kernels' own times: a pair of events per launch is not free (a step of a dozen launches
    paid a tenth of a millisecond for them), and the step is what the end-to-end figures are made of."""
    global LAST_WINDOW
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    w0 = time.time()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    LAST_WINDOW = (w0, time.time())
    ctx.profile(True)
    ctx.profile_read()
    for _ in range(steps):
        r = fn()
    torch.cuda.synchronize()
    prof = ctx.profile_read()
    ctx.profile(False)
    return dt, prof, r


def extra_configs(qe, ctx, torch, dev, k, a, b, c, n, args):
    """The other single-GPU configurations, each on a prefix of the resident table."""
    C = qe.ColumnExpression
    res = {}

    def entry(name, rows, dt, prof, kernel, bpr, launches_per_step, extra=None, steps=5):
        kernel, kk = dominant_kernel(prof, kernel)
        avg = kk["ms"] / kk["launches"] if kk["launches"] else float("nan")
        # (the kernel's time per step, whatever the number of its launches: several batches may share one launch, and a
        # many-groups aggregate has a short trial launch of the same kernel in front)
        gbs = bpr * rows / (kk["ms"] / steps * 1e-3) / 1e9 if kk["launches"] else float("nan")
        e2e = bpr * rows / dt / 1e9
        # frac_hbm_peak_end_to_end: algorithmic bytes / whole step (every kernel, every host wait) - the figure to judge a
        # multi-kernel configuration by; frac_hbm_peak_main_kernel: the same bytes over the dominant kernel's time alone
        e = {"rows": rows, "rows_per_s": rows / dt, "ms_per_step": dt * 1e3, "bytes_per_row": bpr,
             "end_to_end_GBps": e2e, "frac_hbm_peak_end_to_end": e2e / HBM_PEAK_GBS,
             "kernel": kernel, "avg_launch_ms": avg, "kernel_GBps": gbs, "frac_hbm_peak_main_kernel": gbs / HBM_PEAK_GBS,
             "kernels_ms_per_step": {kk: vv["ms"] / steps for kk, vv in prof.items()}}
        if extra:
            e.update(extra)
        if SAMPLER is not None and LAST_WINDOW is not None:
            w = SAMPLER.window(LAST_WINDOW[0], LAST_WINDOW[1], SAMPLER_PCI)
            e["clocks_during_timed_steps"] = {k: v for k, v in w.items() if k not in ("source", "samples_total")}
        res[name] = e

    # The headline plan under the library's default flags (every aggregate waits for its group count and hands back
    # exactly the groups, the reference's protocol, physical_plan/mod.rs:375-488): the deferred mode above is an option.
    hb = [qe.Batch.from_torch(ctx, [t[s:min(n, s + args.batch_rows)] for t in (k, a, b, c)]) for s in range(0, n, args.batch_rows)]
    keyed = lambda col: [qe.SumExpression(C(col)), qe.CountExpression(None)]

    def headline_default():
        sel = qe.SelectionExec([qe.ScanExec(ctx, hb)], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
        proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
        return list(qe.AggregateExec([proj], [C(0)], keyed(1), group_capacity_hint=max(64, 2 * args.groups)).execute())[0]
    dt, prof, out = timed(ctx, torch, headline_default, 5, 2)
    entry("C5_headline_default_flags", n, dt, prof, "qe_aggregate", BYTES_PER_ROW["c5"], len(hb),
          {"groups": out.num_rows, "host_sync": "per aggregate (group count read back, exact-size output)"})
    del out

    # The same plan with order-independent sums (QE_FLAG_DETERMINISTIC_SUM: same bits run after run): an extra pass over
    # the value columns per batch finds the scale, four integer accumulators replace the float one.  What it costs.
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_DETERMINISTIC_SUM | qe.FLAG_JIT_SYNC)
    dt, prof, out = timed(ctx, torch, headline_default, 5, 2)

    def sums_by_key(batch):
        # the aggregate's rows come in table order, which changes from run to run (as the reference's HashMap order does,
        # physical_plan/mod.rs:375-376): the witness compares each group's sum, so the rows are put in key order first
        cols = batch.columns()
        order = np.argsort(np.asarray(cols[0].values), kind="stable")
        return np.asarray(cols[0].values)[order].tobytes() + np.asarray(cols[1].values)[order].tobytes()
    sums_1 = sums_by_key(out)
    sums_2 = sums_by_key(headline_default())
    entry("C5_headline_deterministic_sum", n, dt, prof, "qe_aggregate", BYTES_PER_ROW["c5"], len(hb),
          {"groups": out.num_rows, "same_bits_on_a_second_run": sums_1 == sums_2,
           "host_sync": "one for the scale (a pass over the first 2^22 rows of the first batch) and one per aggregate (the group "
                        "count, with the word the kernels set when a later value does not fit that scale)"})
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP)
    del out

    # A shape no kernel was built for ahead of time: key in column 2, predicate c < 0.3, value a*b - c.  The run-time
    # compiler specialises it (QE_FLAG_JIT_SYNC: compile on first use, here during the warm-up steps).
    rj = min(n, 500_000_000)
    bj = [qe.Batch.from_torch(ctx, [t[s:min(rj, s + args.batch_rows)] for t in (a, b, k, c)]) for s in range(0, rj, args.batch_rows)]
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC)

    def jit_shape():
        sel = qe.SelectionExec([qe.ScanExec(ctx, bj)], qe.LtExpression(C(3), qe.LiteralFloatExpression(0.3)))
        proj = qe.ProjectionExec([sel], [C(2), qe.SubExpression(qe.MulExpression(C(0), C(1)), C(3))])
        return list(qe.AggregateExec([proj], [C(0)], keyed(1), group_capacity_hint=64).execute())[0]
    dt, prof, out = timed(ctx, torch, jit_shape, 5, 2)
    entry("C5_shape_not_prebuilt_jit_500M", rj, dt, prof, "qe_aggregate", BYTES_PER_ROW["c5"], len(bj),
          {"groups": out.num_rows, "plan": "Selection(c < 0.3) -> Projection(k, a*b - c) -> Aggregate(k; SUM, COUNT), columns (a, b, k, c)"})
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP)
    del out, bj, hb

    # C2: fused filter + project, 100 M rows, 50 % selectivity -> materialised, order-preserving output
    r2 = min(n, 100_000_000)
    b2 = [qe.Batch.from_torch(ctx, [a[:r2], b[:r2], c[:r2]])]

    def c2():
        sel = qe.SelectionExec([qe.ScanExec(ctx, b2)], qe.GtExpression(C(2), qe.LiteralFloatExpression(0.5)))
        return list(qe.ProjectionExec([sel], [qe.AddExpression(C(0), qe.MulExpression(C(1), C(2)))]).execute())[0]
    # QE_FLAG_DEFERRED_ROW_COUNT: the batch comes back with the kernel queued and its row count on the way to pinned memory
    # (read when somebody asks: here after the timed steps), so the host queues step after step the way the reference's
    # SelectionIterator streams batches (physical_plan/mod.rs:205-229)
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_DEFERRED_ROW_COUNT)
    dt, prof, out = timed(ctx, torch, c2, 5, 2)
    entry("C2_filter_project_100M", r2, dt, prof, "qe_filter", BYTES_PER_ROW["c2"], 1,
          {"rows_out": out.num_rows, "host_sync": "none inside a step (deferred row count)"})
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP)
    del out
    # the same under the default flags: every batch's row count is read back before it is handed over
    dt, prof, out = timed(ctx, torch, c2, 5, 2)
    entry("C2_filter_project_100M_default_flags", r2, dt, prof, "qe_filter", BYTES_PER_ROW["c2"], 1,
          {"rows_out": out.num_rows, "host_sync": "per batch (row count read back)"})
    del out

    # C3: SUM(fare), COUNT(*) GROUP BY passenger_count-style key, 500 M rows (key column + one float64)
    r3 = min(n, 500_000_000)
    b3 = [qe.Batch.from_torch(ctx, [k[s:min(r3, s + args.batch_rows)], c[s:min(r3, s + args.batch_rows)]])
          for s in range(0, r3, args.batch_rows)]

    def c3():
        return list(qe.AggregateExec([qe.ScanExec(ctx, b3)], [C(0)], [qe.SumExpression(C(1)), qe.CountExpression(None)],
                                     group_capacity_hint=64).execute())[0]
    dt, prof, out = timed(ctx, torch, c3, 5, 2)
    entry("C3_groupby_lowcard_500M", r3, dt, prof, "qe_aggregate", BYTES_PER_ROW["c3"], len(b3), {"groups": out.num_rows})

    # C3 with an Int32 key (the reference's native integer width): 12 B/row
    k32 = k[:r3].to(torch.int32)
    b3i = [qe.Batch.from_torch(ctx, [k32[s:min(r3, s + args.batch_rows)], c[s:min(r3, s + args.batch_rows)]])
           for s in range(0, r3, args.batch_rows)]

    def c3i():
        return list(qe.AggregateExec([qe.ScanExec(ctx, b3i)], [C(0)], [qe.SumExpression(C(1)), qe.CountExpression(None)],
                                     group_capacity_hint=64).execute())[0]
    dt, prof, out = timed(ctx, torch, c3i, 5, 2)
    entry("C3_groupby_lowcard_int32key_500M", r3, dt, prof, "qe_aggregate", 12, len(b3i), {"groups": out.num_rows})
    del k32, b3i

    # C5's plan with 1 M uniformly distributed groups (high-cardinality mode; nearly every row misses the workgroups'
    # tables, so after a look at the first rows the first pass partitions its rows itself: no log, no scatter pass)
    r5 = min(n, 500_000_000)
    uk = torch.empty(r5, dtype=torch.int64, device=dev)
    ctx.generate(1, 99, 0, r5, 0, 1_000_000, uk.data_ptr())
    ctx.synchronize()
    b5 = [qe.Batch.from_torch(ctx, [uk[s:min(r5, s + args.batch_rows)], a[s:min(r5, s + args.batch_rows)],
                                    b[s:min(r5, s + args.batch_rows)], c[s:min(r5, s + args.batch_rows)]])
          for s in range(0, r5, args.batch_rows)]

    def c5m():
        sel = qe.SelectionExec([qe.ScanExec(ctx, b5)], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
        proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
        return list(qe.AggregateExec([proj], [C(0)], [qe.SumExpression(C(1)), qe.CountExpression(None)],
                                     group_capacity_hint=1_100_000).execute())[0]
    dt, prof, out = timed(ctx, torch, c5m, 3, 1)
    entry("C5_plan_1M_groups_500M", r5, dt, prof, "qe_aggregate_direct_static", BYTES_PER_ROW["c5"], len(b5),
          {"groups": out.num_rows, "note": "first pass + partition-wise aggregate: see kernels_ms_per_step"}, steps=3)
    del uk, b5, out

    # C4: 1 M distinct zipf(1.1) Int64 keys, SUM / MIN / MAX / AVG of one float64 column
    r4 = min(n, 500_000_000)
    zk = torch.empty(r4, dtype=torch.int64, device=dev)
    ctx.generate(2, 77, 0, r4, 1_000_000, 1.1, zk.data_ptr())
    ctx.synchronize()
    b4 = [qe.Batch.from_torch(ctx, [zk[s:min(r4, s + args.batch_rows)], c[s:min(r4, s + args.batch_rows)]])
          for s in range(0, r4, args.batch_rows)]

    def c4():
        return list(qe.AggregateExec([qe.ScanExec(ctx, b4)], [C(0)],
                                     [qe.SumExpression(C(1)), qe.MinExpression(C(1)), qe.MaxExpression(C(1)), qe.AvgExpression(C(1))],
                                     group_capacity_hint=1_100_000).execute())[0]
    dt, prof, out = timed(ctx, torch, c4, 2, 1)
    entry("C4_groupby_zipf_1Mkeys_500M", r4, dt, prof, "qe_aggregate_spill_static", BYTES_PER_ROW["c4"], len(b4),
          {"groups": out.num_rows, "note": "not HBM bound: see kernels_ms_per_step"}, steps=2)
    del zk, b4, out

    # The one workload the reference's own tests pin: test_filter (tests/mod.rs:138-188) - Scan -> Selection(gender ==
    # "Female"), all 13 columns of userdata.parquet kept (10 of them Utf8) - at 100 M rows: the fixture's 1000 rows
    # repeated, in batches of 10 M rows (Utf8Array<i32> offsets bound a batch's bytes per column to 2 GiB).
    # (QE_FLAG_JIT_SYNC: the predicate's kernel is compiled during the warm-up step; deferred row count: no batch waits for
    # the device, a batch's rows are read when the batch two after it has been queued)
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC | qe.FLAG_DEFERRED_ROW_COUNT)
    try:
        res["utf8_test_filter_shape_100M"] = utf8_filter_config(qe, ctx, torch, dev)
        res["utf8_test_filter_shape_100M"]["host_sync"] = "none per batch (deferred row count: rows and byte totals read two batches later)"
        ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP | qe.FLAG_JIT_SYNC)
        res["utf8_test_filter_shape_100M_default_flags"] = utf8_filter_config(qe, ctx, torch, dev)
        res["utf8_test_filter_shape_100M_default_flags"]["host_sync"] = "two per batch (row count, byte totals)"
    except FileNotFoundError as e:
        res["utf8_test_filter_shape_100M"] = {"skipped": str(e)}
    ctx.set_flags(qe.FLAG_REFERENCE_SCALAR_ARRAY_SWAP)

    # The headline plan read straight off a Parquet file (uncompressed, PLAIN Float64 + dictionary Int64 key; 16 M
    # rows in row groups of 8 M): host reads the page cache into pinned memory, the GPU decodes.  Host-link bound.
    try:
        if SAMPLER is not None:
            SAMPLER.pause(True)
        res["parquet_scan_headline_plan_16M"] = parquet_config(qe, ctx, torch, k, a, b, c, min(n, 16_000_000))
    except ImportError as e:  # pyarrow writes the file
        res["parquet_scan_headline_plan_16M"] = {"skipped": str(e)}
    finally:
        if SAMPLER is not None:
            SAMPLER.pause(False)
    return res


USERDATA_COLUMNS = ["registration_dttm", "id", "first_name", "last_name", "email", "gender", "ip_address", "cc", "country", "birthdate",
                    "salary", "title", "comments"]


def utf8_filter_config(qe, ctx, torch, dev, repeats=10_000, n_batches=10):
    z = np.load(os.path.join(ROOT, "tests", "golden", "userdata.npz"), allow_pickle=False)
    base_rows = 1000
    rows = base_rows * repeats
    cols, keep, in_bytes = [], [], 0
    for name in USERDATA_COLUMNS:
        valid = z[name + ".valid"]
        d = {"length": rows}
        if not valid.all():
            bits = torch.from_numpy(np.packbits(valid, bitorder="little")).to(dev).repeat(repeats)  # 1000 rows = 125 whole bytes
            pad = torch.zeros(64, dtype=torch.uint8, device=dev)
            bits = torch.cat([bits, pad])
            keep.append(bits)
            d["validity"], d["null_count"] = bits.data_ptr(), int((~valid).sum()) * repeats
            in_bytes += rows // 8
        if name + ".offsets" in z:
            offs, data = z[name + ".offsets"].astype(np.int64), z[name + ".bytes"]
            total = int(offs[-1])
            o = (torch.from_numpy(offs[:-1]).to(dev)[None, :] + total * torch.arange(repeats, device=dev, dtype=torch.int64)[:, None]).reshape(-1)
            o = torch.cat([o, torch.tensor([total * repeats], device=dev, dtype=torch.int64)]).to(torch.int32)
            b = torch.cat([torch.from_numpy(data[:total].copy()).to(dev).repeat(repeats), torch.zeros(64, dtype=torch.uint8, device=dev)])
            keep += [o, b]
            d.update(type=qe.TYPE_UTF8, offsets=o.data_ptr(), values=b.data_ptr(), values_bytes=total * repeats)
            in_bytes += 4 * (rows + 1) + total * repeats
        else:
            v = z[name + ".values"]
            t = torch.from_numpy(v).to(dev).repeat(repeats)
            keep.append(t)
            ty = qe.TYPE_TIMESTAMP_NS if name == "registration_dttm" else {np.dtype(np.int32): qe.TYPE_INT32, np.dtype(np.int64): qe.TYPE_INT64,
                                                                            np.dtype(np.float64): qe.TYPE_FLOAT64}[v.dtype]
            d.update(type=ty, values=t.data_ptr())
            in_bytes += v.dtype.itemsize * rows
        cols.append(d)
    torch.cuda.synchronize()
    batches = [qe.Batch.from_device_columns(ctx, cols, keepalive=keep) for _ in range(n_batches)]  # the same buffers, scanned n_batches times
    gender = USERDATA_COLUMNS.index("gender")

    def plan(with_bytes=False):
        sel = qe.SelectionExec([qe.ScanExec(ctx, batches)], qe.EqExpression(qe.ColumnExpression(gender), qe.LiteralStringExpression("Female")))
        from collections import deque
        n_out, out_bytes, window = 0, 0, deque()

        def account(out):
            nonlocal n_out, out_bytes
            n_out += out.num_rows
            # (a column's descriptor carries its null count, counted on demand by a kernel the host waits for: the sizes
            # of the output buffers are therefore taken in a pass of their own, outside the timed steps)
            for i in range(out.num_columns if with_bytes else 0):
                dsc = out.column_desc(i)
                out_bytes += dsc.values_bytes + (4 * (dsc.length + 1) if dsc.offsets else 0) + (dsc.length // 8 if dsc.validity else 0)
        for out in sel.execute():  # one output batch per input batch; a batch is looked at (and dropped) two batches later
            window.append(out)
            if len(window) > 2:
                account(window.popleft())
        while window:
            account(window.popleft())
        return n_out, out_bytes
    dt, prof, (n_out, _) = timed(ctx, torch, plan, 2, 1)
    _, out_bytes = plan(with_bytes=True)
    total_rows = rows * n_batches
    moved = in_bytes * n_batches + out_bytes
    return {"rows": total_rows, "rows_out": n_out, "columns": 13, "utf8_columns": 10, "rows_per_s": total_rows / dt, "ms_per_step": dt * 1e3,
            "bytes_per_row": moved / total_rows, "end_to_end_GBps": moved / dt / 1e9, "frac_hbm_peak_end_to_end": moved / dt / 1e9 / HBM_PEAK_GBS,
            "kernels_ms_per_step": {kk: vv["ms"] / 2 for kk, vv in prof.items()},
            "note": "algorithmic bytes = every input buffer (values, offsets, validity) read once + every output buffer written once; "
                    "the predicate runs in a kernel compiled for the shape at run time when the context allows (QE_JIT), else in the general kernel"}


def parquet_config(qe, ctx, torch, k, a, b, c, rows):
    import tempfile
    import pyarrow as pa
    import pyarrow.parquet as pq
    C = qe.ColumnExpression
    tbl = pa.table({name: pa.array(t[:rows].cpu().numpy()) for name, t in (("k", k), ("a", a), ("b", b), ("c", c))})
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "headline.parquet")
        pq.write_table(tbl, path, compression="none", use_dictionary=["k"], row_group_size=8_000_000)
        size = os.path.getsize(path)
        # (the file was written a moment ago: let its pages reach the disk now, not during the timed scans - a scan
        # that shares the memory system with their writeback ran a fifth slower for the first half dozen passes)
        fd = os.open(path, os.O_RDONLY)
        try:
            os.fsync(fd)
        finally:
            os.close(fd)

        def plan():
            sel = qe.SelectionExec([qe.ParquetScanExec(ctx, path)], qe.GtExpression(C(3), qe.LiteralFloatExpression(0.5)))
            proj = qe.ProjectionExec([sel], [C(0), qe.AddExpression(C(1), qe.MulExpression(C(2), C(3)))])
            return list(qe.AggregateExec([proj], [C(0)], [qe.SumExpression(C(1)), qe.CountExpression(None)],
                                         group_capacity_hint=64).execute())[0]
        def scan_only():
            # a consumer that takes a row group at a time and lets go of it (the blocks of a batch that is kept are
            # not there for the next scan to reuse: one scan in five would pay for fresh device allocations)
            got = 0
            for batch in qe.ParquetScanExec(ctx, path).execute():
                got += batch.num_rows
            assert got == rows
        # (the first passes over a fresh file run on reader threads, pinned buffers and CPU clocks that are still coming up)
        # (the first half dozen scans of a fresh file run slow - 30, 11, 14, 9, 14 ms, then 8.6-9.1: tools/experiments/parquet_jitter.py)
        dt_scan, prof_scan, _ = timed(ctx, torch, scan_only, 8, 6)
        dt, prof, out = timed(ctx, torch, plan, 8, 2)
        t0 = time.perf_counter()
        pq.read_table(path)
        dt_arrow = time.perf_counter() - t0
    decode_ms = sum(v["ms"] for kk, v in prof_scan.items() if kk.startswith("qe_parquet")) / 8
    return {"rows": rows, "file_bytes": size, "rows_per_s": rows / dt, "ms_per_step": dt * 1e3, "groups": out.num_rows,
            "scan_only_ms": dt_scan * 1e3, "scan_only_file_GBps": size / dt_scan / 1e9, "scan_only_decoded_GBps": rows * 32 / dt_scan / 1e9,
            "decode_kernels_ms_per_scan": decode_ms,
            "pyarrow_read_table_ms": dt_arrow * 1e3,
            "note": "end to end from the page cache; bound by the host link (pinned copy + PCIe), the decode kernels are a "
                    "few percent of it; pyarrow.read_table on the host cores is an Arrow sibling, not the reference's reader"}


def cpu_baseline(rows, seed):
    """The reference-shaped CPU path (oracle/qe_oracle.c, one materialising pass per expression node,
    filter_chunk over every column, single-pass hash aggregate) on a bounded sample of the same synthetic
    table.  `value` is one thread (the reference is single threaded); `all_cores` shards the rows over the
    host threads this box gives one GPU job.  Reported beside the GPU number; never part of `value`."""
    import ctypes as Ct
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    L = orc.lib()
    n = rows
    p = lambda x: x.ctypes.data_as(Ct.c_void_p)
    k = np.empty(n, np.int64)
    a = np.empty(n, np.float64)
    b = np.empty(n, np.float64)
    c = np.empty(n, np.float64)
    L.orc_generate(3, Ct.c_uint64(seed + 1), Ct.c_int64(0), Ct.c_int64(n), Ct.c_double(0), Ct.c_double(0), p(k))
    L.orc_generate(0, Ct.c_uint64(seed + 2), Ct.c_int64(0), Ct.c_int64(n), Ct.c_double(-1), Ct.c_double(1), p(a))
    L.orc_generate(0, Ct.c_uint64(seed + 3), Ct.c_int64(0), Ct.c_int64(n), Ct.c_double(-1), Ct.c_double(1), p(b))
    L.orc_generate(0, Ct.c_uint64(seed + 4), Ct.c_int64(0), Ct.c_int64(n), Ct.c_double(0), Ct.c_double(1), p(c))
    half = np.array([0.5])
    g = 64

    def shard_buffers(lo, hi):
        m0 = hi - lo
        return dict(mask=np.zeros((m0 + 7) // 8 + 8, np.uint8), fk=np.empty(m0, np.int64), fa=np.empty(m0, np.float64),
                    fb=np.empty(m0, np.float64), fc=np.empty(m0, np.float64), t1=np.empty(m0, np.float64), v=np.empty(m0, np.float64))

    def shard_pass(lo, hi, w):
        """All passes over rows [lo, hi) into the preallocated buffers `w`: returns the shard's (keys, row counts, sums)."""
        m0 = hi - lo
        ks, as_, bs, cs = k[lo:hi], a[lo:hi], b[lo:hi], c[lo:hi]
        ok, orows, onn = np.zeros(g, np.uint64), np.zeros(g, np.int64), np.zeros(g, np.int64)
        osum = np.zeros(g, np.float64)
        L.orc_cmp_f64(3, p(cs), p(half), Ct.c_int64(m0), 1, p(w["mask"]))                                 # c > 0.5
        m = 0
        for src, dst in ((ks, w["fk"]), (as_, w["fa"]), (bs, w["fb"]), (cs, w["fc"])):                    # filter_chunk: every column
            m = L.orc_filter_fixed(8, p(src), None, Ct.c_int64(m0), p(w["mask"]), None, p(dst), None)
        L.orc_arith_f64(11, p(w["fb"]), p(w["fc"]), Ct.c_int64(m), 0, p(w["t1"]))                         # b * c
        L.orc_arith_f64(9, p(w["fa"]), p(w["t1"]), Ct.c_int64(m), 0, p(w["v"]))                           # a + (b*c)
        ng = L.orc_hash_aggregate(p(w["fk"].view(np.uint64)), None, None, Ct.c_int64(m), 3, p(w["v"]), None, Ct.c_int64(g), p(ok),
                                  p(orows), p(onn), p(osum), None, None, None, None, None)
        return ok, orows, osum, ng

    whole = shard_buffers(0, n)
    shard_pass(0, n, whole)  # touch every page once, untimed
    reps, t_total = 0, 0.0
    while t_total < 10.0 and reps < 8:
        t0 = time.perf_counter()
        shard_pass(0, n, whole)
        t_total += time.perf_counter() - t0
        reps += 1
    out = {"value": n * reps / t_total, "unit": "rows/s", "cores": 1, "kind": "port",
           "sample": f"{reps} x {n} rows of the same synthetic table (seed {seed}), reference-shaped passes "
                     "(cmp_scalar, filter x4 columns, mul, add, hash aggregate) in oracle/qe_oracle.c, -O3 -march=native"}
    threads = max(1, min(16, os.cpu_count() or 1))
    del whole
    bounds = [(i * n // threads, (i + 1) * n // threads) for i in range(threads)]
    bufs = [shard_buffers(lo, hi) for lo, hi in bounds]
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(lambda i: shard_pass(*bounds[i], bufs[i]), range(threads)))  # untimed warm-up
        reps2, t2 = 0, 0.0
        while t2 < 5.0 and reps2 < 8:
            t0 = time.perf_counter()
            parts = list(pool.map(lambda i: shard_pass(*bounds[i], bufs[i]), range(threads)))
            # merge of the shard tables (a few dozen rows: numpy, not a Python loop, so that it weighs what it would in C)
            keys = np.concatenate([ok[:int(ng)] for ok, _, _, ng in parts])
            uk, inv = np.unique(keys, return_inverse=True)
            merged_rows = np.bincount(inv, weights=np.concatenate([orows[:int(ng)] for _, orows, _, ng in parts]).astype(np.float64), minlength=len(uk))
            merged_sums = np.bincount(inv, weights=np.concatenate([osum[:int(ng)] for _, _, osum, ng in parts]), minlength=len(uk))
            t2 += time.perf_counter() - t0
            reps2 += 1
    out["all_cores"] = {"value": n * reps2 / t2, "unit": "rows/s", "cores": threads,
                        "sample": f"{reps2} x {n} rows, row ranges over {threads} threads (ctypes releases the GIL), shard tables merged"}
    out["reference_literal_aggregate"] = literal_aggregate_baseline(orc, k, c)
    return out


def literal_aggregate_baseline(orc, k, v, rows=1_000_000):
    """The reference's own aggregate loop (physical_plan/mod.rs:326-364: per distinct hash of a batch one membership mask
    over the whole batch and one masked reduction per accumulator, O(rows x groups)), restated in oracle/qe_oracle.c, on
    1 M rows of the synthetic table: MAX and MIN (the only aggregates the reference lowers, query_planner.rs:90-100) of a
    float64 column, grouped by the 10-value key and by a 1000-value key derived from it.  One thread, one batch."""
    res = {"rows": rows, "unit": "rows/s", "cores": 1, "kind": "port"}
    rows = min(rows, len(k))
    Cx = orc.ColumnExpression
    for label, keys in (("groups_10", k[:rows]), ("groups_1000", (k[:rows] * 100 + (np.arange(rows) % 100)).astype(np.int64))):
        chunk = orc.Chunk([orc.Array(orc.INT64, np.ascontiguousarray(keys)), orc.Array(orc.FLOAT64, np.ascontiguousarray(v[:rows]))])
        reps, total, groups = 0, 0.0, 0
        while total < 3.0 and reps < 5:
            t0 = time.perf_counter()
            got = list(orc.AggregateExec([orc.ScanExec([chunk])], [Cx(0)], [orc.MaxExpression(Cx(1)), orc.MinExpression(Cx(1))], literal=True).execute())[0]
            total += time.perf_counter() - t0
            reps += 1
            groups = len(got.arrays[0].values)
        res[label] = {"value": rows * reps / total, "groups": groups, "seconds_per_pass": total / reps}
    return res


if __name__ == "__main__":
    main()
