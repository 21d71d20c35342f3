"""the N > 1 bench holds the line of its first candidate (single-step driver, RCCL send/recv) and
prints it whatever a later candidate does.  bench.candidate_loop is driven here by two gloo ranks on the CPU with the
slab drivers on the oracle-backed engine and stub candidates that raise, disagree, stall past their budget or never
return; every run must end with exit code 0 and ONE valid line with n_gpus = 2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "bench_failsoft_worker.py")


def _run(stubs, timeout=240):
    code = ("import bench, sys; sys.exit(bench.launch_ranks(2, [%r], worker=%r, visible=2))" % (stubs, WORKER))
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def _check_line(lines):
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0 and line["unit"] == "MLUPS"
    assert len(line["batches_ms_per_step"]) == 2
    # the N > 1 line is self-sufficient -- the CPU baseline rank 0 measured, the roofline object
    # with its traffic field (null here: no PMC pass of a stub workload), and the devices RCCL's ranks sat on
    assert line["cpu_baseline"] == {"value": 1.0, "unit": "MLUPS", "cores": 1, "kind": "port", "sample": "stub"}
    assert "traffic" in line["roofline"] and line["roofline"]["traffic"] is None
    seen = line["config"]["transport"]["ranks_seen"]
    assert seen["ranks"] == 2 and seen["distinct_devices"] == 2 and len(seen["device_ids_sha256_8"]) == 2
    return line, line["config"]["transport"]


def test_a_good_second_candidate_may_replace_the_held_line():
    r, lines = _run("rccl")
    assert r.returncode == 0, r.stderr[-3000:]
    line, t = _check_line(lines)
    assert set(t["warmup_ms_per_step"]) == {"single-step/rccl", "two-step/rccl"}
    assert "bit-identical" in t["checks"]["two-step/rccl"] and "after the timed batches too" in t["checks"]["two-step/rccl"]
    assert t["failures"] == {}
    assert t["chosen"] in ("single-step/rccl", "two-step/rccl")
    assert len(t["rank_checksums"]) == 2 and all(abs(c - 8 * 4 * 8) < 1e-6 for c in t["rank_checksums"])   # mass of a rank's slab


@pytest.mark.parametrize("stub,expect", [("raises-in-build", "unavailable"), ("raises-in-batch", "failed in the timed batches"),
                                         ("stalls", "over its wall budget")])
def test_a_failing_candidate_leaves_the_held_line(stub, expect):
    r, lines = _run(stub + ",rccl")
    assert r.returncode == 0, r.stderr[-3000:]
    line, t = _check_line(lines)
    assert expect in t["failures"][f"two-step/{stub}"]
    # the candidate AFTER the failing one still ran and was checked
    assert "bit-identical" in t["checks"]["two-step/rccl"]
    assert t["chosen"] in ("single-step/rccl", "two-step/rccl")


def test_a_candidate_that_disagrees_is_rejected():
    r, lines = _run("disagrees")
    assert r.returncode == 0, r.stderr[-3000:]
    line, t = _check_line(lines)
    assert t["chosen"] == "single-step/rccl"
    assert "MISMATCH" in t["checks"]["two-step/disagrees"]


def test_a_candidate_that_never_returns_ends_in_the_held_line():
    r, lines = _run("never-returns")
    assert r.returncode == 0, r.stderr[-3000:]
    line, t = _check_line(lines)
    assert t["chosen"] == "single-step/rccl"
    assert "watchdog" in t["aborted"]


def test_a_reference_candidate_that_never_returns_ends_in_a_line_that_says_so():
    """the first candidate hangs (RCCL's first point-to-point call on a node, say): every rank ends itself after the
    first candidate's budget and rank 0 prints a line with value 0 and the reason -- no hang until the caller's limit"""
    r, lines = _run("first-never-returns", timeout=120)
    assert r.returncode != 0
    assert len(lines) == 1, (lines, r.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] == 0.0 and line["cpu_baseline"]["kind"] == "port"
    t = line["config"]["transport"]
    assert t["ranks_seen"]["distinct_devices"] == 2
    assert t["chosen"] is None and "watchdog" in t["aborted"] and "single-step/never-returns" in t["aborted"]
