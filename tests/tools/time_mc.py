"""Launch time of one stream's Monte-Carlo step (the estimators' default: 25 dropout samples of ONE window, T = 6), HIP events:
python tests/tools/time_mc.py [pocket|watch|uarm] [n_samples] [T] [auto|tile16|cluster]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
name = sys.argv[1] if len(sys.argv) > 1 else "pocket"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 25
T = int(sys.argv[3]) if len(sys.argv) > 3 else 6
kern = sys.argv[4] if len(sys.argv) > 4 else "auto"
cfg = orc.MODEL_CONFIGS[name]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0); m.load_state_dict(sd); m.set_kernel(kern)
x = torch.randn(1, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
lib = _hip.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
FL = _hip.FLAG_BROADCAST_X | _hip.FLAG_DROPOUT_PHILOX
def run(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, FL, None, 0.2, 1234 + i, C.c_void_p(y.data_ptr()), st), "fwd")
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
run(50)
v = [run(100) for _ in range(9)]
m.check()
print(f"{name} MC samples={B} T={T} kernel={kern} ({m.kernel_name(B, T)}): median {np.median(v):7.2f} us  min {min(v):7.2f}")
