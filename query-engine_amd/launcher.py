"""One process per GPU: starts the ranks of a single-node job as fresh child processes.

The reference is a single-threaded pull iterator (physical_plan/mod.rs:58) and has no launcher; this
one exists so that `bench.py --gpus N` really runs N ranks when nothing (torchrun, a scheduler) has
set WORLD_SIZE for it.  The children are started BEFORE the parent touches a GPU: a process that has
initialised the device must never be replaced or forked into ranks.  Needs neither torch nor a GPU
itself, so the launcher is testable on a CPU box with gloo."""
import os
import socket
import subprocess
import sys
import time


class LaunchError(RuntimeError):
    pass


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environment(rank, world, port, base=None):
    """The environment of one rank: what torch.distributed's env:// rendezvous reads."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    # dmabuf IPC is the only kind the host driver supports: RCCL between processes needs it
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def check_world(requested, env=None):
    """`--gpus N` against what a launcher put into the environment: (world, rank, local_rank, launched).
    launched = False means no launcher ran and the caller has to start the ranks itself when N > 1.
    A WORLD_SIZE that differs from N is an error: the number printed would not be the one asked for."""
    env = os.environ if env is None else env
    if "WORLD_SIZE" not in env:
        return requested, 0, 0, False
    world = int(env["WORLD_SIZE"])
    if world != requested:
        raise LaunchError(f"--gpus {requested} but the launcher started WORLD_SIZE={world} ranks")
    return world, int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", env.get("RANK", "0"))), True


def _tail(path, limit=4000):
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(max(0, size - limit))
            return f.read().decode(errors="replace")
    except OSError:
        return ""


def spawn_ranks(argv, world, timeout=None, env=None, poll=0.2, stderr_dir=None):
    """Runs `argv` once per rank (rank r gets RANK = LOCAL_RANK = r) and waits for all of them.
    Rank 0 inherits stdout, so whatever it prints is the job's output.  Every rank inherits stderr, unless
    `stderr_dir` names a directory: then rank r writes it to <stderr_dir>/rank_<r>.stderr, and when the job
    fails or runs out of time the end of every rank's file is copied to this process's stderr under a line that
    says which ranks failed or were still running - a rendezvous that never completes then ends with a message
    and a non-zero code instead of silence.
    Returns 0 when every rank exited with 0.  When one rank fails or the time limit passes (code 124), the others
    are terminated (the exact processes started here, by pid) and the first failing code is returned."""
    if world < 1:
        raise LaunchError("world size must be at least 1")
    port = free_port()
    procs, logs = [], []
    if stderr_dir:
        os.makedirs(stderr_dir, exist_ok=True)
    try:
        for r in range(world):
            log = open(os.path.join(stderr_dir, f"rank_{r}.stderr"), "wb") if stderr_dir else None
            logs.append(log)
            procs.append(subprocess.Popen(list(argv), env=rank_environment(r, world, port, env),
                                          stdout=None if r == 0 else subprocess.DEVNULL, stderr=log))
        deadline = None if timeout is None else time.monotonic() + timeout
        failed, why = 0, ""
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = bad[0][1]
                why = "rank(s) " + ", ".join(f"{r} (exit code {c})" for r, c in bad) + " failed"
                break
            if all(c == 0 for c in codes):
                return 0
            if deadline is not None and time.monotonic() > deadline:
                failed = 124
                running = [str(r) for r, c in enumerate(codes) if c is None]
                why = f"rank(s) {', '.join(running)} still running after {timeout:.0f} s (a rendezvous or collective that never completed?)"
                break
            time.sleep(poll)
        print(f"launcher: {why}; stopping the other ranks", file=sys.stderr, flush=True)
        return failed
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for r, log in enumerate(logs):
            if log is None:
                continue
            log.close()
            if any(p.returncode != 0 for p in procs):
                text = _tail(log.name)
                if text.strip():
                    print(f"---- rank {r} stderr (end of {log.name}) ----\n{text}", file=sys.stderr, flush=True)


def launch_self(world, gpus_available, need_gpus=True, timeout=None):
    """Re-runs the current script (sys.argv) as `world` ranks.  `gpus_available` must come from a call that
    does not initialise the device (torch.cuda.device_count() on this stack).  Returns the exit code."""
    if need_gpus and gpus_available < world:
        raise LaunchError(f"{world} ranks need {world} GPUs, this machine has {gpus_available}: refusing to run fewer ranks than asked for")
    return spawn_ranks([sys.executable] + sys.argv, world, timeout=timeout)
