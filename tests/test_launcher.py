"""`bench.py --gpus N` must run N ranks or fail loudly; the launcher that starts them is exercised here on CPU
with gloo (the reference has no counterpart: it is a single-threaded pull iterator, physical_plan/mod.rs:58)."""
import importlib.util
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launcher():
    spec = importlib.util.spec_from_file_location("qe_launcher", os.path.join(ROOT, "query-engine_amd", "launcher.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


CHILD = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(dist.get_rank() + 1)])
    dist.all_reduce(t)
    seen = [None] * dist.get_world_size()
    dist.all_gather_object(seen, {"rank": int(os.environ["RANK"]), "local": int(os.environ["LOCAL_RANK"]), "pid": os.getpid()})
    if dist.get_rank() == 0:
        print(json.dumps({"world": dist.get_world_size(), "sum": float(t.item()), "seen": seen}), flush=True)
    dist.destroy_process_group()
    sys.exit(int(os.environ.get("FAIL_RANK", "-1")) == int(os.environ["RANK"]))
""")


def _run_parent(tmp_path, world, extra_env=None):
    """A parent process that spawns `world` gloo ranks through spawn_ranks; returns (exit code, rank 0's stdout)."""
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    parent = tmp_path / "parent.py"
    parent.write_text(textwrap.dedent(f"""
        import importlib.util, sys
        spec = importlib.util.spec_from_file_location("qe_launcher", {os.path.join(ROOT, "query-engine_amd", "launcher.py")!r})
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        sys.exit(m.spawn_ranks([sys.executable, {str(child)!r}], {world}, timeout=120))
    """))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, str(parent)], env=env, capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout


def test_spawn_ranks_runs_every_rank_once(tmp_path):
    code, out = _run_parent(tmp_path, 2)
    assert code == 0, out
    line = json.loads(out.strip().splitlines()[-1])
    assert line["world"] == 2 and line["sum"] == 3.0
    assert sorted(s["rank"] for s in line["seen"]) == [0, 1] and sorted(s["local"] for s in line["seen"]) == [0, 1]
    assert len({s["pid"] for s in line["seen"]}) == 2  # two processes, not one rank counted twice


def test_spawn_ranks_reports_a_failing_rank(tmp_path):
    code, _ = _run_parent(tmp_path, 2, {"FAIL_RANK": "1"})
    assert code != 0


def test_check_world():
    m = _launcher()
    assert m.check_world(1, {}) == (1, 0, 0, False)
    assert m.check_world(4, {}) == (4, 0, 0, False)  # nobody launched: the caller starts the ranks
    assert m.check_world(2, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}) == (2, 1, 1, True)
    with pytest.raises(m.LaunchError):
        m.check_world(8, {"WORLD_SIZE": "2", "RANK": "0"})
    env = m.rank_environment(3, 4, 12345, base={})
    assert env["RANK"] == "3" and env["WORLD_SIZE"] == "4" and env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "12345"
    with pytest.raises(m.LaunchError):
        m.launch_self(2, gpus_available=1)


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` on a machine with fewer than N GPUs exits non-zero with a message: never a silent N = 1."""
    import torch
    have = torch.cuda.device_count()
    want = have + 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "QE_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert f"needs {want} GPUs" in r.stderr
    # a launcher that started a different number of ranks than --gpus names is an error too
    env["WORLD_SIZE"], env["RANK"] = str(want + 1), "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
