"""Launch time of ImuPoseLSTM (input layer + 2 x 256 LSTM + head), HIP events: python tests/tools/time_imupose.py [B] [T] [auto|tile16]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kern = sys.argv[3] if len(sys.argv) > 3 else "auto"
FLAGS = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0       # diagnostic bits, e.g. 0x02000000: any-placement clusters
m = nn_models.ImuPoseLSTM(22, 256, 2, 14, device=0); m.load_state_dict(orc.make_imupose_state_dict(22, 14, 0)); m.set_kernel(kern)
x = torch.randn(B, T, 22, device="cuda"); y = torch.empty(B, 14, device="cuda"); lib = _hip.lib()
run = lambda: _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, FLAGS, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
for _ in range(15): run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
n = 20
for _ in range(n): run()
b.record(); b.synchronize(); m.check()
us = a.elapsed_time(b) / n * 1e3
flop = B * (T * (2 * 256 * 22 + 2 * 4 * 256 * (512 + 512)) + 2 * 14 * 256)
print(f"ImuPoseLSTM B={B} T={T} kernel={kern} ({m.kernel_name(B, T)}): {us:.1f} us per call, {B / us:.3f} M windows/s, {flop / us / 1e6:.1f} TFLOP/s")
