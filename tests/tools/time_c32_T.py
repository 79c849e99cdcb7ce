"""ape_lstm_cluster32 (or, with f16, ape_lstm_cluster_f16v2) per launch over the window length: python tests/tools/time_c32_T.py [f16] [T ...]"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
args = sys.argv[1:]
f16 = bool(args) and args[0] == "f16"
if f16: args = args[1:]
Ts = [int(a) for a in args] or [1, 2, 3, 6, 8, 16, 64]
name = "watch" if f16 else "pocket"
cfg = orc.MODEL_CONFIGS[name]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
m.set_norm_stats(np.zeros(cfg["I"]), np.ones(cfg["I"]), np.zeros(cfg["O"]), np.ones(cfg["O"]))
if f16: m.set_precision("f16")
lib = _hip.lib()
B = 1024
for T in Ts:
    x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
    def fwd():
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
    for _ in range(30): fwd()
    meds = []
    for blk in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40): fwd()
        b.record(); b.synchronize()
        meds.append(a.elapsed_time(b) / 40 * 1e3)
    m.check()
    print(f"{name} 1024 x {T:3d} {m.last_kernel():32s} {statistics.median(meds):8.2f} us per launch (min block {min(meds):.2f})", flush=True)
