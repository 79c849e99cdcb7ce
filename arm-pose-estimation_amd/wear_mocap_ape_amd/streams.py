"""Data-parallel sharding of independent sensor streams / windows over the GPUs of one node.

The reference has no parallelism of any kind (last row); windows are fully
independent (zero initial state per window, nn_models.py:180-189), so the path shards with NO
per-step exchange: one process per GPU, a contiguous range of streams per rank, and a single
broadcast of the weight blob from rank 0 at start-up (RCCL over xGMI when the process group's
backend is "nccl"; the same code runs on "gloo" for CPU tests)."""
from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_streams: int, rank: int, world_size: int) -> Tuple[int, int]:
    """contiguous [lo, hi) of stream indices owned by ``rank``; sizes differ by at most one and
    the ranges tile [0, n_streams) in rank order (8192 streams on 8 GPUs -> 1024 each)."""
    if n_streams < 0 or world_size < 1 or not (0 <= rank < world_size):
        raise UserWarning(f"bad shard request: n={n_streams} rank={rank} world={world_size}")
    base, rem = divmod(n_streams, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def flatten_state_dict(state_dict, keys) -> np.ndarray:
    """state_dict -> flat float32 blob in ``keys`` order (the layout ``ape_model_load_weights`` takes)"""
    parts = []
    for k in keys:
        v = state_dict[k]
        a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        parts.append(np.ascontiguousarray(a, dtype=np.float32).reshape(-1))
    return np.concatenate(parts)


def broadcast_blob(blob, n_floats: int, device: torch.device, src: int = 0) -> torch.Tensor:
    """rank ``src`` passes the float32 blob, the others pass None; every rank returns a tensor on
    ``device`` holding identical bytes.  One collective, init-time only."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return torch.as_tensor(blob, dtype=torch.float32).to(device)
    # the collective runs where the backend lives: device memory for nccl (RCCL over xGMI), host for gloo
    comm_dev = device if dist.get_backend() == "nccl" else torch.device("cpu")
    if dist.get_rank() == src:
        t = torch.as_tensor(blob, dtype=torch.float32).to(comm_dev).contiguous()
        if t.numel() != n_floats:
            raise UserWarning(f"blob has {t.numel()} floats, expected {n_floats}")
    else:
        t = torch.empty((n_floats,), dtype=torch.float32, device=comm_dev)
    dist.broadcast(t, src=src)
    return t.to(device)


def broadcast_stats(stats: dict, I: int, O: int, device: torch.device, src: int = 0) -> dict:
    """float64 normalisation statistics travel the same way (exact bits)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return stats
    comm_dev = device if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.empty((2 * I + 2 * O,), dtype=torch.float64, device=comm_dev)
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.concatenate([np.asarray(stats[k], dtype=np.float64).reshape(-1)
                                                 for k in ("xx_m", "xx_s", "yy_m", "yy_s")])))
    dist.broadcast(t, src=src)
    h = t.cpu().numpy()
    return {"xx_m": h[:I], "xx_s": h[I:2 * I], "yy_m": h[2 * I:2 * I + O], "yy_s": h[2 * I + O:]}
