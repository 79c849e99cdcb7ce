"""Median launch time of the cluster LSTM kernel alone (HIP events), for A/B runs of build variants:
APE_HIP_LIB=<lib> python tests/tools/time_cluster.py [pocket|watch|uarm] [B] [T]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models

name = sys.argv[1] if len(sys.argv) > 1 else "pocket"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = orc.MODEL_CONFIGS[name]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0); m.load_state_dict(sd)
x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
lib = _hip.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
if os.environ.get('APE_PRECISION') == 'f16': m.set_precision('f16')
lib.ape_model_set_kernel(m.handle, {'auto': 0, 'tile16': 1, 'cluster': 2, 'cluster_gen1': 3}[os.environ.get('APE_KERNEL', 'cluster')])
def run(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, 0, None, 0.0, 0, C.c_void_p(y.data_ptr()), st), "fwd")
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
run(20)
v = [run(20) for _ in range(9)]
pick = np.unique(np.r_[0:4, np.linspace(0, B - 1, 24).astype(int), max(0, B - 4):B])      # rows from every part of the batch
ref = orc.lstm_forward(sd, x.cpu().numpy()[pick])[:, -1]
err = float(np.abs(y.cpu().numpy()[pick] - ref).max())
m.check()
flop = m.flops_per_window(T) * B
print(f"{os.environ.get('APE_HIP_LIB', 'default'):20s} {m.kernel_name(B, T):34s} {name} B={B} T={T}: median {np.median(v):8.1f} us  min {min(v):8.1f}  {flop / np.median(v) / 1e6:6.1f} TFLOP/s  max|dy| {err:.1e}")
