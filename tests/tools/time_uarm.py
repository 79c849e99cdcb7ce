"""Launch time of the third deployed regressor, WatchPhoneUarmNN's 3 x 128 LSTM (watch_phone_uarm_nn.py:13-41: I = 38, O = 12), HIP
events: python tests/tools/time_uarm.py [B] [kernel]   (T = 6 as deployed, and T = 64)"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = orc.MODEL_CONFIGS["uarm"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0))
if len(sys.argv) > 2: m.set_kernel(sys.argv[2])
lib = _hip.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
FLAGS = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0       # diagnostic bits, e.g. 0x01000000: the one-workgroup-per-CU form of ape_lstm_cluster16
for T in (tuple(int(v) for v in sys.argv[3].split(',')) if len(sys.argv) > 3 else (6, 64)):
    x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
    run = lambda: _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, FLAGS, None, 0.0, 0, C.c_void_p(y.data_ptr()), st), "fwd")
    for _ in range(30): run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(100): run()
    b.record(); b.synchronize()
    us = a.elapsed_time(b) / 100 * 1e3
    print(f"uarm (38,128,3,12) B={B} T={T} kernel={m.kernel_name(B, T)}: {us:.1f} us per launch, {B / us:.3f} M windows/s, "
          f"{m.flops_per_window(T) * B / us / 1e6:.1f} TFLOP/s = {m.flops_per_window(T) * B / us / 1e6 / 157.3 * 100:.1f} % of the f32 MFMA peak")
m.check()
