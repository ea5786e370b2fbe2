"""bench.py --gpus N launches its own ranks when it is not already one: rehearsed on the CPU with gloo and a
stand-in step (`--rehearse`), which exercises the launcher, the process group, the barrier-bracketed timed region and the
MAX-over-ranks reduction - everything but the kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.pop("LOCAL_RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse", "--steps", "5", "--warmup", "1"] + extra,
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_ranks_are_self_launched():
    line = _run(["--gpus", "2"])
    assert line["n_gpus"] == 2 and line["world_size_seen_by_backend"] == 2
    assert line["steps"] == 5 and line["warmup"] == 1 and line["ms_per_step"] >= 1.0
    assert line["scaling"] == "weak" and line["rehearsal"] is True


def test_single_rank_runs_in_process():
    line = _run([])
    assert line["n_gpus"] == 1 and line["world_size_seen_by_backend"] == 1
