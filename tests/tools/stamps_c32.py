"""Per-section cycle accounting of lstm_cluster32.hip (diagnostic build, make diag): steady-state sections of cluster 0 / member 0, per wave and
layer: top wait, barrier, MFMA spans (with the exchange hooks), everything behind the barrier (spans + drain + cell update + publish).
python tests/tools/stamps_c32.py [B] [T] [extra flags, e.g. 0x00400000 = the opt-in plain in-XCD hand-over]"""
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
cfg = orc.MODEL_CONFIGS["pocket"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0))
m.set_kernel("cluster")
x = torch.randn(B, T, cfg["I"], device="cuda")
EXTRA = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
lib = _hip.lib()
y = torch.empty(B, cfg["O"], device="cuda")
for _ in range(5):
    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, EXTRA, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
torch.cuda.synchronize(); m.check()
buf = (C.c_ulonglong * 2048)()
lib.ape_debug_read_wg.argtypes = [C.c_void_p, C.c_void_p]
lib.ape_debug_read_wg(m.handle, buf)
d = np.frombuffer(buf, dtype=np.uint64)[32:32 + 64].reshape(4, 16)[:, :10].reshape(4, 2, 5).astype(np.float64)
print(f"{m.kernel_name(B, T)}  B={B} T={T} flags {EXTRA:#x}: shader cycles per steady-state section; MFMA content: layer 0 {(4 + 32) * 4 * 64}, layer 1 {64 * 4 * 64}")
for w in range(4):
    for l in range(2):
        n = max(d[w, l, 4], 1)
        top, bar, span, behind = (d[w, l, k] / n for k in range(4))
        print(f"  wave {w} layer {l}: top wait {top:6.1f}  barrier {bar:6.1f}  spans {span:8.1f}  drain + cell update + publish {behind - span:7.1f}  total {top + bar + behind:8.1f}   ({int(n)} sections)")

h = np.frombuffer(buf, dtype=np.uint64)[128:128 + 64].reshape(4, 16)[:, :12].reshape(4, 2, 6).astype(np.float64)
print("inside the spans (cycles per section, each figure includes one stamp pair ~ 2 x s_memtime + LDS drain):")
for w in range(4):
    for l in range(2):
        n = max(d[w, l, 4], 1)
        print(f"  wave {w} layer {l}: store drain + flag {h[w, l, 0] / n:6.1f}  x staging {h[w, l, 1] / n:6.1f}  judge {h[w, l, 2] / n:6.1f}  gather issue (8 pieces) {h[w, l, 3] / n:6.1f}"
              f"  eight hook-free k-blocks {h[w, l, 5] / n:7.1f} (2048 of MFMA)")
