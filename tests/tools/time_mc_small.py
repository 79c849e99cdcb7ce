"""The Monte-Carlo latency kernel (lstm_mc_small.hip): kernel time by HIP events for one window x n samples, the device-side
frame of a one-stream bank, and the estimators' consumer loop (process_row, host in / host out).
python tests/tools/time_mc_small.py [name]"""
import ctypes as C
import sys
import time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
name = sys.argv[1] if len(sys.argv) > 1 else "pocket"
cfg = orc.MODEL_CONFIGS[name]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(sd)
rng = np.random.default_rng(0)
I, O, T = cfg["I"], cfg["O"], cfg["T"]
m.set_norm_stats(rng.normal(size=I), 1 + rng.random(I), rng.normal(size=O) * 0.1, 1 + 0.1 * rng.random(O))
lib = _hip.lib()
x = torch.from_numpy(rng.normal(size=(1, T, I)).astype(np.float32)).cuda()
for kernel in ("auto", "auto_gen1"):
    m.set_kernel(kernel)
    for n in (1, 4, 25, 32, 50, 60, 64, 128):
        y = torch.empty((n, O), dtype=torch.float32, device="cuda")
        us = []
        for i in range(220):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), n, T, _hip.FLAG_DROPOUT_PHILOX | _hip.FLAG_BROADCAST_X,
                                            None, 0.2, 1000 + i, C.c_void_p(y.data_ptr()), None), "fwd")
            b.record(); b.synchronize()
            if i >= 20: us.append(a.elapsed_time(b) * 1e3)
        print(f"{name} {kernel:9s} n={n:3d}: {m.last_kernel():22s} launch p50 {np.percentile(us, 50):6.1f} us  p99 {np.percentile(us, 99):6.1f}")
    m.check()
m.set_kernel("auto")
kind = {"pocket": _hip.PARSE_WATCH_PHONE_POCKET, "watch": _hip.PARSE_WATCH_ONLY, "uarm": _hip.PARSE_WATCH_PHONE_UARM}[name]
width = _hip.PARSE_SHAPES[kind][0]
rows = [torch.from_numpy(rng.normal(size=(1, width)).astype(np.float32)).cuda() for _ in range(4)]
for n_mc, smooth in ((1, 1), (25, 1), (50, 1), (60, 5), (25, 10)):
    bank = StreamBank(m, 1, T, smooth=smooth, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc, dropout=0.2)
    us = []
    for i in range(320):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        bank.push_rows(rows[i % 4], kind)
        bank.step_datagrams()
        b.record(); b.synchronize()
        if i >= 20: us.append(a.elapsed_time(b) * 1e3)
    m.check()
    print(f"{name} bank S=1 n_mc={n_mc} smooth={smooth}: {m.last_kernel()} frame p50 {np.percentile(us, 50):.1f} us  p99 {np.percentile(us, 99):.1f} us")
    # host frame: ape_streams_frame_host
    row_h = rng.normal(size=(width,)).astype(np.float32)
    out_h = np.empty((25 + 6 * smooth * n_mc,), dtype=np.float64)
    us = []
    for i in range(520):
        t0 = time.perf_counter()
        _hip.check(lib.ape_streams_frame_host(bank._handle, kind, C.c_void_p(row_h.ctypes.data), _hip.FLAG_NORMALIZE_INPUT,
                                              C.c_void_p(out_h.ctypes.data), _hip.F64, None), "frame_host")
        if i >= 20: us.append((time.perf_counter() - t0) * 1e6)
    print(f"    ape_streams_frame_host p50 {np.percentile(us, 50):.1f} us  p99 {np.percentile(us, 99):.1f} us")
    del bank
