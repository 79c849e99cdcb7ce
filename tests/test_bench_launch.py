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


def test_bench_helpers_on_cpu(tmp_path, monkeypatch):
    """pure host helpers of bench.py: the sysfs GPU count never touches HIP, the tail report names the outliers, a traffic figure
    measured on another build of the kernel is flagged stale"""
    import json
    import sys
    sys.path.insert(0, str(REPO))
    import numpy as np
    import bench
    n = bench.visible_gpus_without_opening_them()
    assert n is None or n >= 0
    us = np.full(1000, 17.0); us[[3, 500]] = [40.0, 90.0]
    rep = bench._tail_report(us, "x")
    assert rep["max_us"] == 90.0 and rep["outliers_above_1p5x_median"]["frame_indices"] == [3, 500]
    # staleness: an entry is fresh only if its object hash is the one of the library in lib/build_info.json
    info = json.loads((REPO / "arm-pose-estimation_amd" / "lib" / "build_info.json").read_text())
    obj, sha = next(iter(info["objects"].items()))
    fake = {"kernels": [{"tag": "t_old", "kernel": "k_test", "windows": 7, "hbm_bytes_per_launch": 1.0},
                        {"tag": "t_new", "kernel": "k_test", "windows": 7, "hbm_bytes_per_launch": 2.0, "object": obj, "object_sha256": sha},
                        {"tag": "t_other", "kernel": "k_other", "windows": 7, "hbm_bytes_per_launch": 3.0, "object": obj, "object_sha256": "0" * 64}]}
    prof = tmp_path / "profiles"; prof.mkdir()
    (prof / "traffic_latest.json").write_text(json.dumps(fake))
    (tmp_path / "arm-pose-estimation_amd" / "lib").mkdir(parents=True)
    (tmp_path / "arm-pose-estimation_amd" / "lib" / "build_info.json").write_text(json.dumps(info))
    monkeypatch.setattr(bench, "REPO", tmp_path)
    assert bench.load_traffic("k_test<1>", 7) == (2.0, "t_new", False)        # the newest matching entry, stamped with this build
    assert bench.load_traffic("k_other", 7) == (3.0, "t_other", True)          # measured on another build
    assert bench.load_traffic("k_none", 7) == (None, None, None)
