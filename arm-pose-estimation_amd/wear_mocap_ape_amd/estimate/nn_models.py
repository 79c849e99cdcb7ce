"""HIP-backed pose regressor with the call surface of the reference's ``DropoutLSTM``
(``estimate/nn_models.py:160-207``) and its checkpoint loader (``:373-415``).

The module holds no arithmetic: ``forward`` hands device pointers to ``ape_lstm_forward`` of
``libape_hip.so`` (fused L-layer LSTM + linear head, MFMA f32, state resident on chip).
torch is used for device memory and streams only."""
import json
import logging
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch

import wear_mocap_ape_amd.config as config
from wear_mocap_ape_amd import _hip


class _TrainFlag:
    """stands in for ``self.lstm`` of the reference module: ``model.lstm.train()`` is how
    ``monte_carlo_predictions`` switches inter-layer dropout on -- permanently (nn_models.py:204)."""

    def __init__(self):
        self.training = False

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)


def state_dict_keys(hidden_layer_count: int):
    keys = []
    for k in range(hidden_layer_count):
        keys += [f"lstm.weight_ih_l{k}", f"lstm.weight_hh_l{k}", f"lstm.bias_ih_l{k}", f"lstm.bias_hh_l{k}"]
    return keys + ["output_layer.weight", "output_layer.bias"]


_LAYOUT_FOR_OUTPUTS = {14: _hip.LAYOUT_ORI_CAL_LARM_UARM_HIPS, 12: _hip.LAYOUT_ORI_CAL_LARM_UARM,
                       20: _hip.LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS}


class DropoutLSTM:
    """``DropoutLSTM(input_size, hidden_layer_size, hidden_layer_count, output_size, dropout)``.

    ``model(x)`` with ``x`` float32 ``[B,T,I]`` returns float32 ``[B,T,O]`` (head on every step,
    nn_models.py:188-189); ``model.monte_carlo_predictions(n_samples, x)`` repeats a batch-1 window
    ``n_samples`` times and runs it with inter-layer dropout (nn_models.py:191-207).  Inputs may
    live on the host or on the model's GPU; the result is returned where the input was."""

    _MODEL_KIND = _hip.MODEL_LSTM

    def __init__(self, input_size, hidden_layer_size, hidden_layer_count, output_size, dropout=0.2,
                 device: int = None, target_layout: int = None):
        self.input_size = int(input_size)
        self.hidden_layer_size = int(hidden_layer_size)
        self.hidden_layer_count = int(hidden_layer_count)
        self.output_size = int(output_size)
        self.dropout = float(dropout)
        self.lstm = _TrainFlag()
        self._mc_calls = 0
        self._seed = 0x5EED
        self._state = None
        self._pending = []
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.device_index = int(device)
        if target_layout is None:
            target_layout = _LAYOUT_FOR_OUTPUTS.get(self.output_size, _hip.LAYOUT_NONE)
        self.target_layout = target_layout
        self._create_handle(self._MODEL_KIND)

    def _create_handle(self, model_kind: int):
        import ctypes as C
        self._dims = _hip.ApeDims(self.input_size, self.hidden_layer_size, self.hidden_layer_count,
                                  self.output_size, self.target_layout, self.device_index, model_kind)
        self._handle = C.c_void_p()
        _hip.check(_hip.lib().ape_model_create(C.byref(self._dims), C.byref(self._handle)), "ape_model_create")

    # ---- lifetime -----------------------------------------------------------------------------
    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                _hip.lib().ape_model_destroy(h)
            except Exception:
                pass
            self._handle = None

    @property
    def handle(self):
        return self._handle

    @property
    def torch_device(self):
        return torch.device("cuda", self.device_index)

    # ---- weights ------------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        """same keys and shapes as the reference module's state_dict."""
        want = self._wanted_shapes()
        missing = [k for k in want if k not in state_dict]
        extra = [k for k in state_dict if k not in want]
        if missing or extra:
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {extra}")
        parts, kept = [], OrderedDict()
        for key, shape in want.items():
            v = state_dict[key]
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if tuple(a.shape) != shape:
                raise RuntimeError(f"size mismatch for {key}: {tuple(a.shape)} vs {shape}")
            a = np.ascontiguousarray(a, dtype=np.float32)
            kept[key] = a
            parts.append(a.reshape(-1))
        blob = np.concatenate(parts)
        self.load_weight_blob(blob)
        self._state = kept
        return self

    def _wanted_shapes(self, lstm_in: int = None) -> OrderedDict:
        H, O = self.hidden_layer_size, self.output_size
        I = self.input_size if lstm_in is None else lstm_in
        want = OrderedDict()
        for k in range(self.hidden_layer_count):
            want[f"lstm.weight_ih_l{k}"] = (4 * H, I if k == 0 else H)
            want[f"lstm.weight_hh_l{k}"] = (4 * H, H)
            want[f"lstm.bias_ih_l{k}"] = (4 * H,)
            want[f"lstm.bias_hh_l{k}"] = (4 * H,)
        want["output_layer.weight"] = (O, H)
        want["output_layer.bias"] = (O,)
        return want

    def load_weight_blob(self, blob):
        """flat float32 blob in state_dict order: a host numpy array or a CUDA tensor (e.g. the
        buffer an RCCL broadcast filled)."""
        import ctypes as C
        n = int(_hip.lib().ape_weight_blob_floats(C.byref(self._dims)))
        if isinstance(blob, torch.Tensor):
            if blob.dtype != torch.float32 or blob.numel() != n or not blob.is_contiguous():
                raise UserWarning(f"weight blob must be contiguous float32 with {n} elements")
            ptr = blob.data_ptr()
        else:
            blob = np.ascontiguousarray(blob, dtype=np.float32)
            if blob.size != n:
                raise UserWarning(f"weight blob must have {n} elements, got {blob.size}")
            ptr = blob.ctypes.data
        _hip.check(_hip.lib().ape_model_load_weights(self._handle, C.c_void_p(ptr), n), "ape_model_load_weights")
        self._state = None

    def weight_blob_floats(self) -> int:
        import ctypes as C
        return int(_hip.lib().ape_weight_blob_floats(C.byref(self._dims)))

    def state_dict(self):
        if self._state is None:
            raise UserWarning("no state_dict loaded through load_state_dict")
        return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in self._state.items())

    def eval(self):
        self.lstm.eval()
        return self

    def train(self, mode: bool = True):
        self.lstm.train(mode)
        return self

    # ---- estimator configuration (host pointers, float64) --------------------------------------
    def set_norm_stats(self, xx_m, xx_s, yy_m, yy_s):
        import ctypes as C
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (xx_m, xx_s, yy_m, yy_s)]
        if arrs[0].size != self.input_size or arrs[1].size != self.input_size or \
                arrs[2].size != self.output_size or arrs[3].size != self.output_size:
            raise UserWarning("norm stats do not match the model dimensions")
        _hip.check(_hip.lib().ape_model_set_norm_stats(self._handle, *[_hip.dptr(a, C.c_double) for a in arrs]),
                   "ape_model_set_norm_stats")

    def set_body(self, body_measurements):
        import ctypes as C
        b = np.ascontiguousarray(np.asarray(body_measurements, dtype=np.float64).reshape(-1))
        if b.size != 9:
            raise UserWarning("body measurements must hold 9 values")
        _hip.check(_hip.lib().ape_model_set_body(self._handle, _hip.dptr(b, C.c_double)), "ape_model_set_body")

    # ---- forward --------------------------------------------------------------------------------
    def _mask_shape(self, B, T, last_step_only):
        return (self.hidden_layer_count - 1, B, T, self.hidden_layer_size)

    def _run(self, x, flags, masks=None, dropout_p=0.0, seed=0, last_step_only=False, rows=None, hs=None):
        import ctypes as C
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x), dtype=torch.float32)
        if x.dim() != 3 or x.shape[2] != self.input_size:
            raise UserWarning(f"expected x of shape [B,T,{self.input_size}], got {tuple(x.shape)}")
        if x.shape[0] < 1 or x.shape[1] < 1:
            raise UserWarning(f"empty batch or window: {tuple(x.shape)}")
        on_host = not x.is_cuda
        dev = self.torch_device
        with torch.cuda.device(dev):
            xd = x.to(device=dev, dtype=torch.float32).contiguous()
            B, T = int(xd.shape[0]), int(xd.shape[1])
            if rows is not None:                 # one window shared by `rows` batch rows (MC mode)
                if B != 1:
                    raise UserWarning("MC predictions only for batch size 1")
                B = int(rows)
                flags |= _hip.FLAG_BROADCAST_X
            if last_step_only:
                y = torch.empty((B, 1, self.output_size), dtype=torch.float32, device=dev)
            else:
                y = torch.empty((B, T, self.output_size), dtype=torch.float32, device=dev)
                flags |= _hip.FLAG_ALL_STEPS
            mptr = None
            if masks is not None:
                masks = masks.to(device=dev, dtype=torch.float32).contiguous()
                want = self._mask_shape(B, T, last_step_only)
                if tuple(masks.shape) != want:
                    raise UserWarning(f"masks must have shape {want}, got {tuple(masks.shape)}")
                mptr = C.c_void_p(masks.data_ptr())
                flags |= _hip.FLAG_DROPOUT_MASKS
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if hs is not None:
                # (h_0, c_0), each [L,B,H], as nn.LSTM takes them (nn_models.py:188: self.lstm(x, hs))
                if not isinstance(hs, (tuple, list)) or len(hs) != 2:
                    raise UserWarning("hs must be a tuple (h_0, c_0)")
                want = (self.hidden_layer_count, B, self.hidden_layer_size)
                hc = []
                for a in hs:
                    a = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
                    if tuple(a.shape) != want:
                        raise UserWarning(f"h_0 / c_0 must have shape {want}, got {tuple(a.shape)}")
                    hc.append(a.to(device=dev, dtype=torch.float32).contiguous())
                _hip.check(_hip.lib().ape_lstm_forward_hs(self._handle, C.c_void_p(xd.data_ptr()), B, T, flags, mptr,
                                                          float(dropout_p), int(seed), C.c_void_p(hc[0].data_ptr()),
                                                          C.c_void_p(hc[1].data_ptr()), C.c_void_p(y.data_ptr()), stream),
                           "ape_lstm_forward_hs")
            else:
                _hip.check(_hip.lib().ape_lstm_forward(self._handle, C.c_void_p(xd.data_ptr()), B, T, flags, mptr,
                                                       float(dropout_p), int(seed), C.c_void_p(y.data_ptr()), stream),
                           "ape_lstm_forward")
        # ape_model_recover re-issues every call the handle has journaled since its last check, from the raw device pointers it was
        # given -- so the buffers of a journaled call must stay alive (and out of the caching allocator's hands) until then: the mirror
        # holds references, at most the journal's 64 calls (beyond that the library refuses to replay anyway), dropped by check / recover
        self._pending.append((xd, y, masks, hc if hs is not None else None))
        if len(self._pending) > 64:
            del self._pending[0]
        if on_host:
            # the one place where results reach the host, so the health of the launch is checked here: an aborted
            # weight-stationary launch leaves `y` unwritten, and ape_model_recover then runs the call again on the
            # batch-tile kernel (the inputs are still alive here), so the frame is not lost
            self.recover()
            return y.cpu()
        return y

    def forward(self, x, hs=None, masks=None, last_step_only=False, normalize_input=False, rows=None):
        """``masks`` (float32 ``[L-1,B,T,H]`` of 0 or 1/(1-p)) injects explicit dropout masks;
        ``last_step_only`` returns ``[B,1,O]`` (only the step the estimators consume);
        ``normalize_input`` fuses the f64 z-score of raw features into the load;
        ``hs = (h_0, c_0)``, each ``[L,B,H]``: initial state, as the reference passes it on to ``nn.LSTM``
        (nn_models.py:180-189); ``None`` = zeros."""
        flags = _hip.FLAG_NORMALIZE_INPUT if normalize_input else 0
        if masks is not None:
            return self._run(x, flags, masks=masks, last_step_only=last_step_only, rows=rows, hs=hs)
        if self.lstm.training and self.dropout > 0.0 and self.hidden_layer_count > 1:
            self._mc_calls += 1
            return self._run(x, flags | _hip.FLAG_DROPOUT_PHILOX, dropout_p=self.dropout,
                             seed=(self._seed << 20) + self._mc_calls, last_step_only=last_step_only, rows=rows, hs=hs)
        return self._run(x, flags, last_step_only=last_step_only, rows=rows, hs=hs)

    __call__ = forward

    def manual_seed(self, seed: int):
        self._seed, self._mc_calls = int(seed), 0

    def monte_carlo_predictions(self, n_samples: int, x, hs=None, last_step_only=False):
        if x.shape[0] > 1:
            raise UserWarning("MC predictions only for batch size 1")     # nn_models.py:201-202
        self.lstm.train()                                                  # permanent, as nn_models.py:204
        # x.repeat((n_samples, 1, 1)) of nn_models.py:206 happens inside the kernel (APE_FLAG_BROADCAST_X)
        return self.forward(x, hs, last_step_only=last_step_only, rows=n_samples)

    def set_kernel(self, choice: str = "auto"):
        """'auto' | 'tile16' | 'cluster' | 'cluster_gen1' (cluster kernels of the first generation only) | 'auto_gen1'
        (auto's dispatch without the second-generation kernels): which LSTM kernel ``forward`` launches (A/B runs, tests)"""
        code = {"auto": _hip.KERNEL_AUTO, "tile16": _hip.KERNEL_TILE16, "cluster": _hip.KERNEL_CLUSTER,
                "cluster_gen1": _hip.KERNEL_CLUSTER_GEN1, "auto_gen1": _hip.KERNEL_AUTO_GEN1}[choice]
        _hip.check(_hip.lib().ape_model_set_kernel(self._handle, code), "ape_model_set_kernel")
        return self

    def set_precision(self, precision: str = "f32"):
        """'f32' (exact float32 MFMA, default) | 'f16' (binary16 weights / inputs / hidden state, float32
        accumulate and cell state: BASELINE configs[4]; last-step output without dropout) | 'f16_gen1' (the same
        arithmetic pinned to the first-generation fp16 kernel, for A/B runs)"""
        code = {"f32": _hip.PRECISION_F32, "f16": _hip.PRECISION_F16, "f16_gen1": _hip.PRECISION_F16_GEN1}[precision]
        _hip.check(_hip.lib().ape_model_set_precision(self._handle, code), "ape_model_set_precision")
        return self

    def check(self):
        """blocking health check: raises if a cluster-kernel launch gave up waiting for a peer workgroup (the handle is
        reset; the outputs of the calls since the last check are invalid)"""
        try:
            _hip.check(_hip.lib().ape_model_check(self._handle), "ape_model_check")
        finally:
            self._pending.clear()            # the journal is empty behind a check, whatever it found

    def recover(self):
        """blocking health check that survives an abort (``ape_model_recover``): the calls made since the last check are
        issued again on the kernels that need no co-residency; raises only if that is impossible.  The device inputs of
        those calls must still be alive and unchanged."""
        try:
            _hip.check(_hip.lib().ape_model_recover(self._handle), "ape_model_recover")
        finally:
            self._pending.clear()

    def stats(self) -> dict:
        """{'aborted_checks', 'reissued_calls', 'lost_calls'} of this handle (``ape_model_stats``)"""
        import ctypes as C
        st = _hip.ApeModelStats()
        _hip.check(_hip.lib().ape_model_stats(self._handle, C.byref(st)), "ape_model_stats")
        return {k: int(getattr(st, k)) for k, _ in st._fields_}

    def kernel_name(self, B: int, T: int) -> str:
        return _hip.lib().ape_lstm_kernel_name(self._handle, B, T).decode()

    def last_kernel(self) -> str:
        """the LSTM kernel the newest call on this model launched (``ape_model_last_kernel``)"""
        return _hip.lib().ape_model_last_kernel(self._handle).decode()

    def flops_per_window(self, T: int) -> float:
        import ctypes as C
        return float(_hip.lib().ape_flops_per_window(C.byref(self._dims), T))


class DropoutFF(DropoutLSTM):
    """HIP-backed MLP regressor with the call surface of the reference's ``DropoutFF``
    (``estimate/nn_models.py:313-370``): ``Linear(I,H)``, ``hidden_layer_count`` x ``Linear(H,H)`` (all with
    leaky_relu), dropout, ``Linear(H,O)``, applied to the last axis of ``x``.  ``model(x)`` maps ``[B,T,I]`` to
    ``[B,T,O]`` (``[B,I]`` to ``[B,O]``); ``monte_carlo_predictions`` switches the dropout in front of the output
    layer on -- permanently, like the reference (nn_models.py:367) -- and repeats the rows ``n_samples`` times."""

    def __init__(self, output_size, hidden_layer_size, hidden_layer_count, input_size, dropout=0.2,
                 device: int = None, target_layout: int = None):
        self.input_size = int(input_size)
        self.hidden_layer_size = int(hidden_layer_size)
        self.hidden_layer_count = int(hidden_layer_count)
        self.output_size = int(output_size)
        self.dropout = float(dropout)
        self._do = _TrainFlag()
        self.lstm = self._do                     # shared plumbing looks at `.lstm.training`
        self._mc_calls = 0
        self._seed = 0x5EED
        self._state = None
        self._pending = []
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.device_index = int(device)
        if target_layout is None:
            target_layout = _LAYOUT_FOR_OUTPUTS.get(self.output_size, _hip.LAYOUT_NONE)
        self.target_layout = target_layout
        self._create_handle(_hip.MODEL_FF)

    def state_keys(self):
        keys = ["_input_layer.weight", "_input_layer.bias"]
        for k in range(self.hidden_layer_count):
            keys += [f"_hidden_layers.{k}.weight", f"_hidden_layers.{k}.bias"]
        return keys + ["_output_layer.weight", "_output_layer.bias"]

    def load_state_dict(self, state_dict):
        H, I, O = self.hidden_layer_size, self.input_size, self.output_size
        want = OrderedDict()
        want["_input_layer.weight"], want["_input_layer.bias"] = (H, I), (H,)
        for k in range(self.hidden_layer_count):
            want[f"_hidden_layers.{k}.weight"], want[f"_hidden_layers.{k}.bias"] = (H, H), (H,)
        want["_output_layer.weight"], want["_output_layer.bias"] = (O, H), (O,)
        missing = [k for k in want if k not in state_dict]
        extra = [k for k in state_dict if k not in want]
        if missing or extra:
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {extra}")
        parts, kept = [], OrderedDict()
        for key, shape in want.items():
            v = state_dict[key]
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if tuple(a.shape) != shape:
                raise RuntimeError(f"size mismatch for {key}: {tuple(a.shape)} vs {shape}")
            a = np.ascontiguousarray(a, dtype=np.float32)
            kept[key] = a
            parts.append(a.reshape(-1))
        self.load_weight_blob(np.concatenate(parts))
        self._state = kept
        return self

    def _mask_shape(self, B, T, last_step_only):
        return ((B if last_step_only else B * T), self.hidden_layer_size)

    def forward(self, x, masks=None, last_step_only=False, normalize_input=False):
        """``masks``: float32 ``[rows,H]`` of 0 or 1/(1-p) for the dropout in front of the output layer"""
        two_d = (not isinstance(x, torch.Tensor) and np.asarray(x).ndim == 2) or (isinstance(x, torch.Tensor) and x.dim() == 2)
        if two_d:
            x = x[:, None, :]
        flags = _hip.FLAG_NORMALIZE_INPUT if normalize_input else 0
        if masks is not None:
            y = self._run(x, flags, masks=masks, last_step_only=last_step_only)
        elif self._do.training and self.dropout > 0.0:
            self._mc_calls += 1
            y = self._run(x, flags | _hip.FLAG_DROPOUT_PHILOX, dropout_p=self.dropout,
                          seed=(self._seed << 20) + self._mc_calls, last_step_only=last_step_only)
        else:
            y = self._run(x, flags, last_step_only=last_step_only)
        return y[:, 0, :] if two_d else y

    __call__ = forward

    def monte_carlo_predictions(self, n_samples: int, x, last_step_only=False):
        self._do.train()                                                    # nn_models.py:367
        rep = x.repeat((n_samples, 1, 1)) if isinstance(x, torch.Tensor) else np.tile(np.asarray(x), (n_samples, 1, 1))
        return self.forward(rep, last_step_only=last_step_only)

    def set_kernel(self, choice: str = "auto"):
        """'auto' (the two-stage pipeline for chip-filling eval batches, the tile kernel otherwise) | 'tile16' (tile kernel only)"""
        code = {"auto": _hip.KERNEL_AUTO, "tile16": _hip.KERNEL_TILE16}[choice]
        _hip.check(_hip.lib().ape_model_set_kernel(self._handle, code), "ape_model_set_kernel")
        return self

    def set_precision(self, precision: str = "f32"):
        if precision != "f32":
            raise UserWarning("the MLP regressor is float32 only")
        return self


class ImuPoseLSTM(DropoutLSTM):
    """HIP-backed ``ImuPoseLSTM`` (``estimate/nn_models.py:210-249``): ``Linear(I,256)`` + ReLU in front of a 2 x 256
    LSTM and ``Linear(256,O)``.  Like the reference it keeps, and ignores, ``hidden_layer_size`` /
    ``hidden_layer_count`` (:217-229); its ``monte_carlo_predictions`` is the plain forward of the window it is
    given -- no repeat, no dropout (:246-251)."""
    _MODEL_KIND = _hip.MODEL_IMUPOSE

    def __init__(self, input_size, hidden_layer_size, hidden_layer_count, output_size, dropout=0.2,
                 device: int = None, target_layout: int = None):
        super().__init__(input_size, 256, 2, output_size, dropout, device, target_layout)

    def _wanted_shapes(self, lstm_in: int = None) -> OrderedDict:
        want = OrderedDict()
        want["input_layer.weight"], want["input_layer.bias"] = (256, self.input_size), (256,)
        want.update(super()._wanted_shapes(lstm_in=256))
        return want

    def forward(self, x, hs=None, masks=None, last_step_only=False, normalize_input=False, rows=None):
        if masks is not None:
            raise UserWarning("ImuPoseLSTM has no dropout mode")
        return self._run(x, _hip.FLAG_NORMALIZE_INPUT if normalize_input else 0, last_step_only=last_step_only, rows=rows,
                         hs=hs)

    __call__ = forward

    def monte_carlo_predictions(self, n_samples: int, x, hs=None, last_step_only=False):
        return self.forward(x, None, last_step_only=last_step_only)

    def set_precision(self, precision: str = "f32"):
        if precision != "f32":
            raise UserWarning("ImuPoseLSTM is float32 only")
        return self


def load_deployed_model_from_hash(hash_str: str):
    """``hash`` -> ``(model, params)`` exactly like nn_models.py:373-415: reads
    ``<deploy>/nn/<hash>/results.json`` and ``checkpoint.pt`` (a ``(model_state, optimizer_state)``
    tuple), dispatches on ``params["model"]``, leaves the model in eval mode.  Missing files and
    unknown model names raise ``UserWarning`` like the reference."""
    save_path = Path(config.PATHS["deploy"]) / "nn" / hash_str
    json_path = save_path / "results.json"
    chkpt_path = save_path / "checkpoint.pt"
    if not json_path.exists():
        raise UserWarning(f"no json found {json_path}")
    if not chkpt_path.exists():
        raise UserWarning(f"no checkpoint found {chkpt_path}")
    with open(json_path, "r") as f:
        params = json.load(f)
    if params["model"] == "DropoutLSTM":
        params["model"] = DropoutLSTM
    elif params["model"] == "DropoutFF":
        params["model"] = DropoutFF
    elif params["model"] == "ImuPoseLSTM":
        params["model"] = ImuPoseLSTM
    else:
        raise UserWarning(f"{params['model']} not handled")
    nn_model = params["model"](input_size=len(params["x_inputs_v"]), hidden_layer_size=params["hidden_layer_size"],
                               hidden_layer_count=params["hidden_layer_count"],
                               output_size=len(params["y_targets_v"]), dropout=params["dropout"])
    model_state, _ = torch.load(chkpt_path, map_location="cpu")
    nn_model.load_state_dict(model_state)
    nn_model.eval()
    logging.info("loaded model in eval mode from {}".format(save_path))
    return nn_model, params
