"""bench.py's launch contract (CPU part): `--gpus N` must either run N ranks or fail loudly -- never measure one
GPU and report it as N.  The N-rank run itself needs GPUs; what can be checked here is
that every mismatch is refused before anything is measured."""
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def _run(args, env_extra, drop=("RANK", "WORLD_SIZE", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra)
    return subprocess.run([sys.executable, str(REPO / "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus_flag_must_match_the_launchers_world_size():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert '{"metric"' not in r.stdout
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"RANK": "0", "WORLD_SIZE": "4", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_self_spawn_refuses_more_ranks_than_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        return                                  # a multi-GPU host would really start the ranks: not a CPU-suite job
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr
    assert '{"metric"' not in r.stdout
