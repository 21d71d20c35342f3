"""bench.py --gpus N starts its own N ranks: the launcher path is
driven here on CPU with a stub worker, and the refusal to run a smaller job is checked."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "bench_stub_worker.py")


def _run(code):
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)


def test_launcher_starts_n_ranks_and_relays_rank0_line():
    r = _run("import bench, sys; sys.exit(bench.launch_ranks(2, ['--gpus', '2', '--steps', '4'], "
             f"worker={STUB!r}, visible=2))")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["sum"] == 3.0          # both ranks took part in the all-reduce
    assert line["argv"] == ["--gpus", "2", "--steps", "4"]


def test_launcher_relays_a_failing_rank():
    r = _run("import bench, sys; sys.exit(bench.launch_ranks(2, ['--fail'], "
             f"worker={STUB!r}, visible=2))")
    assert r.returncode != 0
    assert "exited with code" in r.stderr


def test_more_gpus_than_visible_is_refused_not_downgraded():
    """On a box with fewer devices than --gpus the run must fail loudly (this container has none)."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True,
                       text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        return                                                  # a real multi-GPU node: nothing to refuse
    assert r.returncode != 0
    assert "--gpus 2 requested but only" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr
