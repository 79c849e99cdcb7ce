"""Per-part cycle sums of the Monte-Carlo latency kernel (diagnostic library, `make -C csrc diag`), member 0 / wave 0:
APE_HIP_LIB=arm-pose-estimation_amd/lib/diag/libape_hip_diag.so python tests/tools/stamps_mc.py [pocket|watch] [n] [T]"""
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
cfg = orc.MODEL_CONFIGS[name]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0); m.load_state_dict(sd)
x = torch.randn(1, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
lib = _hip.lib()
lib.ape_debug_read_wg.restype, lib.ape_debug_read_wg.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
FL = _hip.FLAG_BROADCAST_X | _hip.FLAG_DROPOUT_PHILOX
acc = []
for it in range(60):
    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, FL, None, 0.2, 7 + it, C.c_void_p(y.data_ptr()), None), "fwd")
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (256 * 8))()
    assert lib.ape_debug_read_wg(m.handle, buf) == 0
    if it >= 20:
        acc.append(np.array(buf[:10], dtype=np.float64))
v = np.median(np.array(acc), axis=0)
P = T + 1
names = ["prologue: weights into registers, LDS zeroed, x_0, XCD rendezvous", "x staging + layer 0 (GEMV, gates, n masks, granule stores)",
         "layer 1 (MFMAs, cell update, granule stores)", "first barrier (activation tile free)", "collect: poll rounds + LDS writes",
         "second barrier", "head"]
mhz = v[8] / v[9] * 100
print(f"{name} n={B} T={T}: kernel (member 0, wave 0) {v[8]:.0f} cycles = {v[9] / 100:.2f} us at {mhz:.0f} MHz; {P} phases")
for k, nm in enumerate(names):
    per = f"{v[k] / P:7.0f} cycles = {v[k] / P / mhz:5.2f} us per phase" if 1 <= k <= 5 else ""
    print(f"  {nm:72s} {v[k]:8.0f} cycles {v[k] / mhz:6.2f} us  {v[k] / v[8] * 100:5.1f} %  {per}")
m.check()
