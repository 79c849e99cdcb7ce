"""Per-section cycle accounting of lstm_cluster16.hip (diagnostic build, make diag): steady-state sections of cluster 0 / member 0, per
wave and layer: span A (+ the deferred cell update), counted wait, barrier, span B.  python tests/tools/stamps_uarm16.py [B] [T]"""
import ctypes as C, os, sys
os.environ.setdefault("APE_HIP_LIB", "/root/repo/arm-pose-estimation_amd/lib/diag/libape_hip_diag.so")
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = orc.MODEL_CONFIGS["uarm"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0))
m.set_kernel("cluster")
x = torch.randn(B, T, cfg["I"], device="cuda")
for _ in range(5): m(x, last_step_only=True)
torch.cuda.synchronize(); m.check()
lib = _hip.lib(); buf = (C.c_ulonglong * 2048)()
lib.ape_debug_read_wg.argtypes = [C.c_void_p, C.c_void_p]
lib.ape_debug_read_wg(m.handle, buf)
d = np.frombuffer(buf, dtype=np.uint64)[:128].reshape(8, 16)[:, :15].reshape(8, 3, 5).astype(np.float64)
print(f"{m.kernel_name(B, T)}  B={B} T={T}: cycles per steady-state section (100 MHz counter x 24 -> shader clocks if constant)")
for w in (0, 4):
    for l in range(3):
        n = max(d[w, l, 4], 1)
        a, wt, bar, b = (d[w, l, k] / n for k in range(4))
        print(f"  wave {w} layer {l}: span A {a:7.1f}  wait {wt:6.1f}  barrier {bar:6.1f}  span B {b:7.1f}  total {a + wt + bar + b:7.1f}   ({int(n)} sections)")
