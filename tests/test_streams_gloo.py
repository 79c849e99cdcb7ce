"""N>1 path on CPU: two processes over gloo exercise the stream sharding and the one start-up
broadcast (the same code runs over RCCL on the GPU node: backend "nccl")."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parents[1]


def test_shard_ranges_tile_the_streams():
    from wear_mocap_ape_amd.streams import shard_range
    for n, world in ((8192, 8), (1024, 1), (10, 4), (3, 8), (0, 2), (1001, 2)):
        spans = [shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))             # contiguous, in rank order
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(8192, 3, 8) == (3072, 4096)                               # 1024 streams per GPU
    with pytest.raises(UserWarning):
        shard_range(10, 4, 4)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    sys.path.insert(0, str(REPO / "arm-pose-estimation_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ape_oracle as orc
    from wear_mocap_ape_amd import streams
    from wear_mocap_ape_amd.estimate.nn_models import state_dict_keys
    cfg = orc.MODEL_CONFIGS["uarm"]
    keys = state_dict_keys(cfg["L"])
    n = 351756
    blob = stats = None
    if rank == 0:
        sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
        blob = streams.flatten_state_dict(sd, keys)
        assert blob.size == n
        rng = np.random.default_rng(0)
        stats = {k: rng.normal(size=(cfg["I"] if k[0] == "x" else cfg["O"])) for k in ("xx_m", "xx_s", "yy_m", "yy_s")}
    got = streams.broadcast_blob(blob, n, torch.device("cpu"))
    st = streams.broadcast_stats(stats, cfg["I"], cfg["O"], torch.device("cpu"))
    lo, hi = streams.shard_range(2048, rank, world)
    np.save(Path(out_dir) / f"blob_{rank}.npy", got.numpy())
    np.save(Path(out_dir) / f"stats_{rank}.npy", np.concatenate([st[k] for k in ("xx_m", "xx_s", "yy_m", "yy_s")]))
    np.save(Path(out_dir) / f"span_{rank}.npy", np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_is_bitwise(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    b0, b1 = np.load(tmp_path / "blob_0.npy"), np.load(tmp_path / "blob_1.npy")
    assert b0.dtype == np.float32 and b0.tobytes() == b1.tobytes()              # identical bytes on every rank
    assert np.load(tmp_path / "stats_0.npy").tobytes() == np.load(tmp_path / "stats_1.npy").tobytes()
    assert np.load(tmp_path / "span_0.npy").tolist() == [0, 1024]
    assert np.load(tmp_path / "span_1.npy").tolist() == [1024, 2048]
