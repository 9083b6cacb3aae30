"""bench.py's own rank start-up (`--gpus N` without a torch.distributed environment), on a CPU box: the ranks come up
through gloo, rendezvous and rank 0's line is relayed; a world size that contradicts --gpus is refused."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_flag_starts_that_many_ranks():
    out = _run(["--gpus", "2", "--selfcheck-launch"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["rank_sum"] == 1


def test_world_size_must_match_gpus_flag():
    out = _run(["--gpus", "2", "--selfcheck-launch"], env={"WORLD_SIZE": "3", "RANK": "0"})
    assert out.returncode != 0
    assert "WORLD_SIZE=3" in out.stderr


def test_single_rank_needs_no_launcher():
    out = _run(["--gpus", "1", "--selfcheck-launch"])
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1
