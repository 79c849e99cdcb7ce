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


def _canned_result(world=1):
    """round 5's full 22.5 KB result (profiles/r05_bench_default.json), widened to `world` ranks"""
    import copy
    import json
    d = json.loads((REPO / "profiles" / "r05_bench_default.json").read_text())
    if world > 1:
        pr = d["config"]["sharding"]["per_rank"][0]
        ranks = []
        for r in range(world):
            q = copy.deepcopy(pr)
            q.update(rank=r, device=r, streams=[1024 * r, 1024 * (r + 1)], bank_S1024_mc25_T6_ms_per_frame=1.1959517161051432,
                     bank_uarm_S1024_mc50_T6_ms_per_frame=1.339531421661377, bank_watch_S1024_mc25_T8_smooth10_ms_per_frame=1.6213542938232421)
            ranks.append(q)
        d["config"]["sharding"].update(ranks=world, backend="nccl", per_rank=ranks)
        d["n_gpus"] = world
        for k in ("batch1", "stream_bank_T6", "other_paths", "dispatch_boundaries", "fp16_config4", "cpu_baseline",
                  "parity_vs_cpu_reference", "gpu_over_cpu"):
            d.pop(k, None)                      # the N-rank run measures the sharded path only
    return d


def test_the_driver_line_stays_small_and_complete():
    """the one stdout line must be strict JSON well under 8 KB and carry roofline + cpu_baseline; the rest of
    what bench.py measures goes to bench_detail.json"""
    import json
    import sys
    sys.path.insert(0, str(REPO))
    import bench
    full = _canned_result()
    assert len(json.dumps(full)) > 20000                      # the input is the line that did not parse
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_CAP_BYTES <= 6144
    d = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))      # no NaN / Infinity
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity_vs_cpu_reference", "gpu_over_cpu"):
        assert k in d, k
    assert d["config"]["workload"].startswith("BASELINE configs[2]") and d["dtype"] == "f32"
    assert "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert rf["traffic"] > 0 and rf["kernel"].startswith("ape_lstm_cluster32")
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "windows/s" and cb["sample"]
    assert len(cb["legs"]) == 4 and all(k.startswith("config3_") for k in cb["legs"])
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-5
    assert abs(d["ms_per_step"] - full["ms_per_step"]) / full["ms_per_step"] < 1e-5
    # one scalar per secondary leg
    assert d["fp16_config4"]["kernel_ms_f16"] > 0 and d["fp16_config4"]["frac"] > 0
    assert d["batch1"]["p50_us"] > 0 and len(d["batch1"]["estimator_loop_p50_p99_us"]) == 3
    assert set(d["bank_frame_ms"]) == set(full["stream_bank_T6"])
    assert d["other_paths_us"]["uarm_lstm_T6"] > 0 and d["other_paths_us"]["kalman_parity"] == "unpinned"
    # a NaN anywhere in the result must not reach the line as a bare NaN token
    full["roofline"]["traffic"] = float("nan")
    assert json.loads(bench.compact_line(full))["roofline"]["traffic"] is None


def test_the_eight_rank_line_stays_under_its_cap():
    import json
    import sys
    sys.path.insert(0, str(REPO))
    import bench
    line = bench.compact_line(_canned_result(world=8))
    d = json.loads(line)
    assert len(line) < 8192 and len(d["config"]["sharding"]["per_rank"]) == 8
    assert all(len(p["bank_ms"]) == 3 and p["bank_ms"][0] > 0 for p in d["config"]["sharding"]["per_rank"])
    assert d["n_gpus"] == 8 and "roofline" in d and "bank_ms_is" in d["config"]["sharding"]


def test_emit_writes_the_detail_beside_the_line(tmp_path, monkeypatch):
    import json
    import os
    import sys
    sys.path.insert(0, str(REPO))
    import bench
    monkeypatch.setattr(bench, "REPO", tmp_path)
    r, w = os.pipe()
    bench.emit(_canned_result(), w)
    os.close(w)
    got = os.read(r, 1 << 16).decode()
    os.close(r)
    assert got.endswith("\n") and got.count("\n") == 1 and len(got) < 6144
    d = json.loads(got)
    assert d["detail"] == "bench_detail.json"
    det = json.loads((tmp_path / "bench_detail.json").read_text())
    assert "dispatch_boundaries" in det and "legs" in det["cpu_baseline"] and len(det["cpu_baseline"]["legs"]) == 16
    assert (tmp_path / "gpurun_out" / "bench_detail.json").exists()
