"""bench.py must start its own ranks when it is called plainly with --gpus N > 1: the driver runs
`python bench.py --gpus N ...` the way it runs `--gpus 1`.  Exercised here without a GPU through --dry-run-launch
(gloo, world size 2): spawn, rendezvous on 127.0.0.1, one collective, ONE JSON line from rank 0, exit code passed on."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=600)


def test_plain_call_with_two_gpus_launches_its_ranks():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-launch"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["rank_sum"] == 3.0 and rec["steps"] == 3


def test_world_size_mismatch_is_an_error():
    out = _run(["--gpus", "2", "--dry-run-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr
