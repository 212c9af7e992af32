"""bench.py's launch contract, the part that can be checked without a GPU: `--gpus N` with no launcher around starts N
ranks of itself only when the box has N devices, and never reports a one-rank run as an N-GPU one."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=300)


def test_refuses_n_ranks_on_fewer_devices():
    r = _run(["--gpus", "2", "--steps", "1"], drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "PRIMME_AMD_BENCH_SHARE_GPU"))
    assert r.returncode != 0
    assert "refusing" in r.stderr and "--gpus 2" in r.stderr
    assert r.stdout.strip() == ""               # no JSON line that could be mistaken for a measurement


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--steps", "1"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert r.stdout.strip() == ""
