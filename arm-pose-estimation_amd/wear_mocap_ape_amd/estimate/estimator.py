"""``Estimator`` base class with the reference's template-method contract
(``estimate/estimator.py:16-218``): window history with pad-by-newest cold start, z-score /
de-normalise bookkeeping in float64, smoothing stack, FK + message, consumer-thread loop.

Host side keeps only bookkeeping (list membership and order -- compared bit-exactly against the
reference in the tests); ``msg_from_pred`` runs the FK and message kernels of libape_hip.so."""
import logging
import queue
import threading
from abc import abstractmethod
from datetime import datetime

import numpy as np
import torch

from wear_mocap_ape_amd.data_types.bone_map import BoneMap
from wear_mocap_ape_amd.estimate import _post
from wear_mocap_ape_amd.utility import data_stats
from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS, TARGET_LAYOUT


class _DeviceFrame:
    """One iteration of the consumer loop (estimator.py:174-177) as ONE call into libape_hip.so: a one-stream bank
    (``ape_streams_*``) keeps the window and smoothing histories on the device, ``ape_streams_frame_host`` takes the raw
    55 / 28-float message (220 / 112 bytes in) and returns the message list ``msg_from_pred`` would (25 + 6N values out) --
    feature builder, z-score, regressor with its Monte-Carlo samples, de-normalisation, smoothing stack, FK and message all run
    on the GPU.  The staged methods of the estimator keep their reference semantics (host histories) for callers that use
    them one by one; this object serves ``processing_loop`` / ``process_row`` only."""

    def __init__(self, model, kind: int, seq_len: int, smooth: int, n_mc: int, normalize: bool, seed: int = 0x5EED):
        import ctypes as C
        from wear_mocap_ape_amd import _hip
        self._C, self._hip, self._lib = C, _hip, _hip.lib()
        self._model, self._kind = model, kind
        self._width = _hip.PARSE_SHAPES[kind][0]
        self._rows = smooth * n_mc
        self._flags = _hip.FLAG_NORMALIZE_INPUT if normalize else 0
        self._bank = C.c_void_p()
        _hip.check(self._lib.ape_streams_create(model.handle, 1, seq_len, smooth, C.byref(self._bank)), "ape_streams_create")
        # monte_carlo_predictions switches the inter-layer dropout on whatever n is (nn_models.py:204), also for one sample
        _hip.check(self._lib.ape_streams_set_mc(self._bank, n_mc, float(model.dropout), int(seed) & (2 ** 64 - 1)),
                   "ape_streams_set_mc")
        self._row = np.empty((self._width,), dtype=np.float32)
        self._out = np.empty((25 + 6 * self._rows,), dtype=np.float64)
        self._row_p = C.c_void_p(self._row.ctypes.data)
        self._out_p = C.c_void_p(self._out.ctypes.data)

    def __del__(self):
        bank, self._bank = getattr(self, "_bank", None), None
        try:
            if bank:
                self._lib.ape_streams_destroy(bank)
        except Exception:          # interpreter shutdown
            pass

    def reset(self):
        self._hip.check(self._lib.ape_streams_reset(self._bank), "ape_streams_reset")

    def frame_stats(self, reset: bool = False) -> dict:
        """where the frames' host time went (ape_streams_frame_stats, ABI 7): per-frame microseconds of the last <= 4096 frames in
        `launch` (rows into pinned staging + the launch calls), `wait` (last launch call returned -> completion words seen) and `copy`
        (pinned output -> the caller's buffer), and the number of frames that fell through to a stream synchronisation"""
        C = self._C

        class _FS(C.Structure):
            _fields_ = [("frames", C.c_uint64), ("fallback_syncs", C.c_uint64), ("recovered", C.c_uint64)]
        fs, n = _FS(), C.c_int32(0)
        trace = np.zeros((4096, 3), dtype=np.float32)
        self._hip.check(self._lib.ape_streams_frame_stats(self._bank, C.byref(fs), C.c_void_p(trace.ctypes.data), 4096, C.byref(n),
                                                          1 if reset else 0), "ape_streams_frame_stats")
        t = trace[:n.value]
        return {"frames": int(fs.frames), "fallback_syncs": int(fs.fallback_syncs), "recovered": int(fs.recovered),
                "launch_us": t[:, 0].copy(), "wait_us": t[:, 1].copy(), "copy_us": t[:, 2].copy()}

    def frame(self, row) -> np.ndarray:
        """raw message -> float64 [25 + 6N]: the message followed by hand / elbow xyz of the N stacked rows (a view of
        this object's buffer, overwritten by the next frame)"""
        self._row[:] = row                       # array('f') (stream/listener/imu.py:68-70), list or ndarray
        self._hip.check(self._lib.ape_streams_frame_host(self._bank, self._kind, self._row_p, self._flags, self._out_p,
                                                         self._hip.F64, None), "ape_streams_frame_host")
        return self._out


class Estimator:
    """Template-method base of the estimators.  Subclasses provide ``parse_row_to_xx`` (raw message ->
    features) and ``make_prediction_from_row_hist`` (normalised window -> NN targets); everything else --
    window and smoothing histories, float64 (de-)normalisation, FK + message, the consumer thread -- lives
    here.  Constructor arguments, methods and properties are the reference's (estimator.py:16-218)."""

    def __init__(self, x_inputs: NNS_INPUTS, y_targets: NNS_TARGETS, normalize: bool = True, smooth: int = 1,
                 seq_len: int = 1, add_mc_samples: bool = True, bonemap: BoneMap = None, tag: str = "Estimator"):
        self.__tag, self._active = tag, False
        self._x_inputs, self._y_targets = x_inputs, y_targets
        self._layout = TARGET_LAYOUT[y_targets]
        self._device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')

        # pre-computed column statistics: float64 arrays xx_m, xx_s [I] and yy_m, yy_s [O]
        self._normalize = normalize
        if normalize:
            self._take_stats(data_stats.get_norm_stats(x_inputs=x_inputs, y_targets=y_targets))

        # histories: feature rows of the current window, predictions of the last `smooth` frames
        self._sequence_len, self._smooth = max(1, seq_len), max(1, smooth)
        self._row_hist, self._smooth_hist = [], []
        self._add_mc_samples, self._last_msg = add_mc_samples, None

        # arm geometry: both arm bones point along -x; [[larm_vec, uarm_vec, uarm_orig_rh]] as one [1,9] row
        larm_len = BoneMap.DEFAULT_LARM_LEN if bonemap is None else bonemap.left_lower_arm_length
        uarm_len = BoneMap.DEFAULT_UARM_LEN if bonemap is None else bonemap.left_upper_arm_length
        self._uarm_orig = BoneMap.DEFAULT_UARM_ORIG_RH if bonemap is None else bonemap.left_upper_arm_origin_rh
        self._larm_vec, self._uarm_vec = np.array([-larm_len, 0, 0]), np.array([-uarm_len, 0, 0])
        self._body_measurements = np.r_[self._larm_vec, self._uarm_vec, self._uarm_orig][np.newaxis, :]
        self._sync_model_config()

    def _take_stats(self, stats: dict):
        self._xx_m, self._xx_s, self._yy_m, self._yy_s = (stats[k] for k in ("xx_m", "xx_s", "yy_m", "yy_s"))

    # subclasses that own a HIP model expose it here so stats/body reach the device handle
    def _hip_model(self):
        return None

    def _sync_model_config(self):
        model = self._hip_model()
        if model is None:
            return
        model.set_body(self._body_measurements)
        if self._normalize:
            model.set_norm_stats(self._xx_m, self._xx_s, self._yy_m, self._yy_s)

    def set_norm_stats(self, stats: dict):
        """replace the statistics loaded at construction (also on the device handle)"""
        self._take_stats(stats)
        self._sync_model_config()
        logging.info("Replaced norm stats xx m+/-s and yy m+/-s")

    def get_last_msg(self):
        return self._last_msg

    def is_active(self):
        return self._active

    def terminate(self):
        self._active = False

    def reset(self):
        self._active, self._row_hist, self._smooth_hist = False, [], []
        frame = getattr(self, "_device_frame", None)
        if frame is not None:
            frame.reset()

    # ---- the device-resident frame (processing_loop's fast path) ----------------------------------------------
    def _frame_samples(self):
        """subclasses with a HIP regressor and a batched feature builder return their Monte-Carlo sample count"""
        return None

    def _frame_runner(self):
        """the one-stream device-side frame of this estimator, or None (no GPU regressor: the staged methods run)"""
        if getattr(self, "_device_frame", None) is None:
            n_mc, model = self._frame_samples(), self._hip_model()
            if n_mc is None or model is None or self._parse_kind is None or not self.use_device_frame:
                return None
            self._device_frame = _DeviceFrame(model, self._parse_kind, self._sequence_len, self._smooth, int(n_mc),
                                              self._normalize)
        return self._device_frame

    use_device_frame = True    # False: processing_loop runs the staged methods (parse -> predict -> message) like the reference
    # True: with add_mc_samples the message is ONE float array of 25 + 6 N values instead of the reference's Python list of them
    # (estimator.py:131-137).  Everything the reference does with the message takes either (`struct.pack('f' * len(msg), *msg)`,
    # stream/publisher/pose_est_udp.py:47; np.array(msg)); building the list is the largest host cost of a Monte-Carlo frame -- 1825 floats
    # at 60 samples x smooth 5: ~20 us of a 58 us frame, and most of its p99 (bench.py batch1.estimator_loop).  Opt-in: the default keeps
    # the reference's type.
    msg_as_array = False

    def process_row(self, row):
        """one iteration of the consumer loop (estimator.py:174-177): raw message -> the message put on the queue"""
        frame = self._frame_runner()
        if frame is None:
            pred = self.add_xx_to_row_hist_and_make_prediction(self.parse_row_to_xx(row))
            return self.msg_from_pred(pred, self._add_mc_samples)
        out = frame.frame(row)
        self._last_msg = out[:25].copy()
        if not self._add_mc_samples:
            return self._last_msg.copy()
        # list of 25 floats followed, for N > 1 stacked rows, by every row's hand and elbow xyz (estimator.py:131-137)
        if self.msg_as_array:
            return out.copy() if out.shape[0] > 31 else self._last_msg.copy()
        return out.tolist() if out.shape[0] > 31 else out[:25].tolist()

    @staticmethod
    def _push_padded(hist: list, item, size: int):
        """append ``item``; a history shorter than ``size`` is filled with copies of the NEWEST item
        (cold start, estimator.py:96-97 / :114-115); the oldest entries beyond ``size`` are dropped"""
        hist.append(item)
        hist.extend([item] * (size - len(hist)))
        del hist[:len(hist) - size]

    def add_xx_to_row_hist_and_make_prediction(self, xx) -> np.array:
        self._push_padded(self._row_hist, xx, self._sequence_len)
        xx_hist = np.vstack(self._row_hist)                       # [T, I]
        if self._normalize:                                       # float64 z-score (estimator.py:103-104)
            xx_hist = (xx_hist - self._xx_m) / self._xx_s
        pred = self.make_prediction_from_row_hist(xx_hist)        # [n, O]
        if self._normalize:                                       # float64 de-normalise (:108-109)
            pred = pred * self._yy_s + self._yy_m
        if self._smooth > 1:                                      # stack of the last `smooth` predictions
            self._push_padded(self._smooth_hist, pred, self._smooth)
            pred = np.vstack(self._smooth_hist)
        return pred

    def msg_from_pred(self, pred: np.array, add_mc_samples: bool) -> np.array:
        # arm_pose_from_nn_targets + msg_from_nn_targets_est in one device round trip
        ctx = _post.context(self._layout)
        with ctx.lock:
            est, msg = _post.fk_and_msg(ctx.handle, self._layout, ctx.device, pred, self._body_measurements)
        self._last_msg = msg.copy()
        if add_mc_samples:
            # list of 25 floats followed, for N > 1 rows, by every row's hand and elbow xyz (estimator.py:131-137)
            msg = list(msg)
            if est.shape[0] > 1:
                msg += list(est[:, :6].reshape(-1))
        return msg

    def process_in_thread(self, sensor_q: queue):
        """start the consumer thread; returns the queue the messages are put on"""
        msg_q = queue.Queue()
        threading.Thread(target=self.processing_loop, args=(sensor_q, msg_q)).start()
        return msg_q

    def _newest_row(self, sensor_q: queue):
        """latency policy of estimator.py:159-161: block up to 2 s for a row; when more than 5 rows are
        queued behind it, skip ahead (newest wins).  Raises ``queue.Empty`` when nothing arrives."""
        row = sensor_q.get(timeout=2)
        while sensor_q.qsize() > 5:
            row = sensor_q.get(timeout=2)
        return row

    def processing_loop(self, sensor_q: queue, msg_q: queue):
        logging.info(f"[{self.__tag}] wearable streaming loop")
        self.reset()
        self._active = True
        tick, frames = datetime.now(), 0          # processing rate is logged every >= 5 s as frames / 5
        while self._active:
            try:
                row = self._newest_row(sensor_q)
            except queue.Empty:
                logging.info(f"[{self.__tag}] no data")
                continue
            now = datetime.now()
            if (now - tick).seconds >= 5:
                logging.info(f"[{self.__tag}] {frames / 5} Hz")
                tick, frames = now, 0
            msg_q.put(self.process_row(row))
            frames += 1

    @abstractmethod
    def make_prediction_from_row_hist(self, xx_hist: np.array) -> np.array:
        return

    @abstractmethod
    def parse_row_to_xx(self, row) -> np.array:
        return

    # ---- batched feature builder: raw messages of many streams -> features on the GPU ----
    _parse_kind = None        # subclasses: the APE_PARSE_* selector of their message/feature layout

    def parse_rows(self, rows, out_dtype=None):
        """rows: float32 ``[N, 55|28]`` raw messages (host array or CUDA tensor) -> features ``[N, I]`` on the
        device, the batched equivalent of ``parse_row_to_xx`` (``ape_parse_rows`` kernel, float64 arithmetic).
        Default dtype is what the reference's ``parse_row_to_xx`` returns (float32; float64 for the
        upper-arm estimator)."""
        import ctypes as C
        from wear_mocap_ape_amd import _hip
        if self._parse_kind is None:
            raise UserWarning("this estimator has no batched feature builder")
        width, n_feat = _hip.PARSE_SHAPES[self._parse_kind]
        if out_dtype is None:
            out_dtype = torch.float64 if self._parse_kind == _hip.PARSE_WATCH_PHONE_UARM else torch.float32
        model = self._hip_model()
        dev = model.torch_device if model is not None else torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(dev):
            rd = torch.as_tensor(rows, dtype=torch.float32).to(dev).contiguous()
            if rd.dim() != 2 or rd.shape[1] != width or rd.shape[0] < 1:
                raise UserWarning(f"expected rows [N>=1,{width}], got {tuple(rd.shape)}")
            xx = torch.empty((rd.shape[0], n_feat), dtype=out_dtype, device=dev)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _hip.check(_hip.lib().ape_parse_rows(self._parse_kind, C.c_void_p(rd.data_ptr()), int(rd.shape[0]),
                                                 C.c_void_p(xx.data_ptr()),
                                                 _hip.F64 if out_dtype == torch.float64 else _hip.F32, stream),
                       "ape_parse_rows")
        return xx

    # ---- batched entry the reference lacks: many independent windows at once ----
    def infer_windows(self, x, est_dtype=torch.float64, return_targets: bool = False):
        """x: raw (un-normalised) features float32 ``[B,T,I]``, host array or CUDA tensor ->
        est ``[B,21|14]`` (same rows ``arm_pose_from_nn_targets`` yields), on the device.
        One ``ape_infer`` call: z-score -> LSTM -> last step -> de-normalise -> FK."""
        import ctypes as C
        from wear_mocap_ape_amd import _hip
        model = self._hip_model()
        if model is None:
            raise UserWarning("this estimator has no HIP regressor")
        dev = model.torch_device
        with torch.cuda.device(dev):
            xd = torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous()
            if xd.dim() != 3 or xd.shape[2] != model.input_size or xd.shape[0] < 1 or xd.shape[1] < 1:
                raise UserWarning(f"expected x [B>=1,T>=1,{model.input_size}], got {tuple(xd.shape)}")
            B, T = int(xd.shape[0]), int(xd.shape[1])
            est = torch.empty((B, _hip.EST_WIDTH[self._layout]), dtype=est_dtype, device=dev)
            y = torch.empty((B, model.output_size), dtype=torch.float32, device=dev) if return_targets else None
            flags = _hip.FLAG_NORMALIZE_INPUT if self._normalize else 0
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _hip.check(_hip.lib().ape_infer(model.handle, C.c_void_p(xd.data_ptr()), B, T, flags,
                                            C.c_void_p(y.data_ptr()) if y is not None else None,
                                            C.c_void_p(est.data_ptr()),
                                            _hip.F64 if est_dtype == torch.float64 else _hip.F32, stream), "ape_infer")
            model._pending.append((xd, y, est))      # journaled by the handle: alive until its next check / recover (nn_models._run)
            if len(model._pending) > 64:
                del model._pending[0]
        return (est, y) if return_targets else est

    # read-only views, same names as the reference's properties (estimator.py:188-218)
    sequence_len = property(lambda self: self._sequence_len)
    body_measurements = property(lambda self: self._body_measurements)
    uarm_orig = property(lambda self: self._uarm_orig)
    uarm_vec = property(lambda self: self._uarm_vec)
    larm_vec = property(lambda self: self._larm_vec)
    device = property(lambda self: self._device)
    x_inputs = property(lambda self: self._x_inputs)
    y_targets = property(lambda self: self._y_targets)
