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


def broadcast_blob(blob, n_floats: int, device: torch.device, src: int = 0, always: bool = False) -> torch.Tensor:
    """rank ``src`` passes the float32 blob, the others pass None; every rank returns a tensor on
    ``device`` holding identical bytes.  One collective, init-time only."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not always):     # `always`: 1-rank rehearsals
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


def broadcast_stats(stats: dict, I: int, O: int, device: torch.device, src: int = 0, always: bool = False) -> dict:
    """float64 normalisation statistics travel the same way (exact bits)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not always):
        return stats
    comm_dev = device if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.empty((2 * I + 2 * O,), dtype=torch.float64, device=comm_dev)
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.concatenate([np.asarray(stats[k], dtype=np.float64).reshape(-1)
                                                 for k in ("xx_m", "xx_s", "yy_m", "yy_s")])))
    dist.broadcast(t, src=src)
    h = t.cpu().numpy()
    return {"xx_m": h[:I], "xx_s": h[I:2 * I], "yy_m": h[2 * I:2 * I + O], "yy_s": h[2 * I + O:]}


class StreamBank:
    """The per-frame step of many independent wearable streams with all state on the device: window rings,
    smoothing stacks, regressor, FK and messages (C ABI ``ape_streams_*``).  For every stream it does what one
    ``Estimator`` does per frame (estimator.py:93-137), so S estimator threads of the reference become three kernel
    launches per frame.  Rank-local: give each rank its ``shard_range`` of streams.

    ``model`` is a HIP-backed ``DropoutLSTM`` (nn_models.py) with weights, norm stats and body set.
    ``monte_carlo_samples=None`` runs the regressor once per stream with deterministic weights; an integer n runs
    it n times per stream and frame with inter-layer dropout ``dropout`` (default: the model's) like
    ``monte_carlo_predictions`` (nn_models.py:191-207), and the stack / tail hold ``smooth * n`` rows per stream."""

    def __init__(self, model, n_streams: int, seq_len: int, smooth: int = 1, normalize: bool = True,
                 dtype: torch.dtype = torch.float32, monte_carlo_samples=None, dropout=None, seed: int = 0x5EED):
        from . import _hip
        import ctypes as C
        self._hip, self._C = _hip, C
        if dtype not in (torch.float32, torch.float64):
            raise UserWarning(f"StreamBank messages are float32 or float64, not {dtype}")
        self._model, self._n, self._smooth = model, n_streams, smooth
        self._flags = _hip.FLAG_NORMALIZE_INPUT if normalize else 0
        self._dtype, self._sel = dtype, (_hip.F32 if dtype == torch.float32 else _hip.F64)
        self._device = torch.device("cuda", model.device_index)
        handle = C.c_void_p()
        _hip.check(_hip.lib().ape_streams_create(model.handle, n_streams, seq_len, smooth, C.byref(handle)),
                   "ape_streams_create")
        self._handle = handle
        self._n_mc = 1
        if monte_carlo_samples is not None:
            self._n_mc = int(monte_carlo_samples)
            p = float(model.dropout if dropout is None else dropout)
            _hip.check(_hip.lib().ape_streams_set_mc(handle, self._n_mc, p, int(seed) & (2 ** 64 - 1)), "ape_streams_set_mc")
        self._msg = torch.empty((n_streams, 25), dtype=dtype, device=self._device)
        self._tail = torch.empty((n_streams, smooth * self._n_mc, 6), dtype=dtype, device=self._device)

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        try:
            if h:
                self._hip.lib().ape_streams_destroy(h)
        except Exception:          # interpreter shutdown: the binding module may already be torn down
            pass

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)

    def reset(self):
        self._hip.check(self._hip.lib().ape_streams_reset(self._handle), "ape_streams_reset")

    def check(self):
        """blocking health check of the model's launches (``ape_model_check``): the bank's outputs stay on the device,
        so the caller decides where to pay for the synchronisation -- e.g. once per batch of frames, before the
        datagrams leave the host"""
        self._model.check()

    def recover(self):
        """like ``check``, but an aborted step is run again on the kernels that need no co-residency (``ape_model_recover``):
        call it behind a step, before the next row is pushed, wherever the datagrams are about to leave the device"""
        self._model.recover()

    def profile(self, enable: bool = True):
        """measurement aid: HIP events around every launch of the step's dominant kernel (``ape_streams_profile``)"""
        self._hip.check(self._hip.lib().ape_streams_profile(self._handle, 1 if enable else 0), "ape_streams_profile")

    def profile_read(self):
        """-> (summed kernel ms, launches) since the last read"""
        ms, n = self._C.c_double(0.0), self._C.c_int32(0)
        self._hip.check(self._hip.lib().ape_streams_profile_read(self._handle, self._C.byref(ms), self._C.byref(n)),
                        "ape_streams_profile_read")
        return float(ms.value), int(n.value)

    def push_rows(self, rows: torch.Tensor, kind: int, big_endian: bool = False):
        """rows: float32 [S, 55|28] on the device -- one raw message per stream (data_types/messaging.py layouts)"""
        width = self._hip.PARSE_SHAPES[kind][0]
        if rows.dtype != torch.float32 or tuple(rows.shape) != (self._n, width) or not rows.is_cuda or not rows.is_contiguous():
            raise UserWarning(f"push_rows wants a contiguous float32 [{self._n},{width}] device tensor")
        k = kind | (self._hip.PARSE_BIG_ENDIAN if big_endian else 0)
        self._hip.check(self._hip.lib().ape_streams_push_rows(self._handle, k, self._C.c_void_p(rows.data_ptr()),
                                                              self._stream()), "ape_streams_push_rows")

    def push_features(self, xx: torch.Tensor):
        """xx: float32 [S, I] on the device -- what ``parse_row_to_xx`` returns, one row per stream"""
        if xx.dtype != torch.float32 or xx.shape[0] != self._n or not xx.is_cuda or not xx.is_contiguous():
            raise UserWarning(f"push_features wants a contiguous float32 [{self._n},I] device tensor")
        self._hip.check(self._hip.lib().ape_streams_push_features(self._handle, self._C.c_void_p(xx.data_ptr()),
                                                                  self._stream()), "ape_streams_push_features")

    def step(self, with_tail: bool = False):
        """-> msg [S,25] (and, with_tail, the hand/elbow xyz of every stacked row [S,smooth*n_mc,6]); the returned
        tensors are the bank's own buffers, overwritten by the next step"""
        tail = self._C.c_void_p(self._tail.data_ptr()) if with_tail else None
        self._hip.check(self._hip.lib().ape_streams_step(self._handle, self._flags, self._C.c_void_p(self._msg.data_ptr()),
                                                         tail, self._sel, self._stream()), "ape_streams_step")
        return (self._msg, self._tail) if with_tail else self._msg

    def step_datagrams(self):
        """-> float32 [S, 25 + 6*smooth*n_mc]: per stream the message followed by the hand/elbow xyz of every stacked
        row -- byte for byte what ``PoseEstPublisherUDP`` sends for one estimator frame (pose_est_udp.py:47 packs
        the list of estimator.py:131-137 as native float32), so ``row.cpu().numpy().tobytes()`` is the datagram.
        Like the reference, rows only carry the tail when there is more than one stacked row."""
        n = self._smooth * self._n_mc
        if n == 1:
            if self._dtype != torch.float32:
                raise UserWarning("step_datagrams wants a float32 bank")
            return self.step()
        if getattr(self, "_packed", None) is None:
            self._packed = torch.empty((self._n, 25 + 6 * n), dtype=torch.float32, device=self._device)
        self._hip.check(self._hip.lib().ape_streams_step(self._handle, self._flags | self._hip.FLAG_PACKED_MSG,
                                                         self._C.c_void_p(self._packed.data_ptr()), None, self._hip.F32,
                                                         self._stream()), "ape_streams_step")
        return self._packed
