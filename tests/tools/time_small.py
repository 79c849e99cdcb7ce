"""Launch time of the small-batch latency kernel (B <= 4), HIP events, with parity against the oracle:
APE_HIP_LIB=<lib> python tests/tools/time_small.py [pocket|watch|uarm] [B] [T]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
name = sys.argv[1] if len(sys.argv) > 1 else "pocket"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cfg = orc.MODEL_CONFIGS[name]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0); m.load_state_dict(sd)
x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
lib = _hip.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
FLAGS = int(os.environ.get("APE_FLAGS", "0"), 0)
def run(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, FLAGS, None, 0.0, 0, C.c_void_p(y.data_ptr()), st), "fwd")
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
run(50)
v = [run(100) for _ in range(9)]
ref = orc.lstm_forward(sd, x.cpu().numpy())[:, -1]
err = float(np.abs(y.cpu().numpy() - ref).max()); m.check()
print(f"{os.environ.get('APE_HIP_LIB', 'default')[-24:]:24s} {name} B={B} T={T}: median {np.median(v):7.2f} us  min {min(v):7.2f}  max|dy| {err:.1e}")
