"""Per-part cycle sums of the Monte-Carlo latency kernel (diagnostic library built by `make -C csrc diag`), cluster 0 / member 0 / wave 0:
APE_HIP_LIB=arm-pose-estimation_amd/lib/diag/libape_hip_diag.so python tests/tools/stamps_mc_small.py [pocket|watch|uarm] [n] [T]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models

name = sys.argv[1] if len(sys.argv) > 1 else "pocket"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
cfg = orc.MODEL_CONFIGS[name]
T = int(sys.argv[3]) if len(sys.argv) > 3 else cfg["T"]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0); m.load_state_dict(sd)
x = torch.randn(1, T, cfg["I"], device="cuda"); y = torch.empty(n, cfg["O"], device="cuda")
lib = _hip.lib()
lib.ape_debug_read_wg.restype, lib.ape_debug_read_wg.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
acc = []
for it in range(60):
    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), n, T, _hip.FLAG_DROPOUT_PHILOX | _hip.FLAG_BROADCAST_X, None, 0.2, it,
                                    C.c_void_p(y.data_ptr()), None), "fwd")
    torch.cuda.synchronize()
    assert m.last_kernel() == "ape_lstm_mc_small"
    buf = (C.c_ulonglong * (256 * 8))()
    assert lib.ape_debug_read_wg(m.handle, buf) == 0
    if it >= 20:
        acc.append(np.array(buf[:16], dtype=np.float64))
vv = np.median(np.array(acc), axis=0)
P = T + cfg["L"] - 1
names = ["prologue: weights into registers, x_0, masks of phase 0, XCD rendezvous", "x staging + layer 0 (GEMV, gates, granule)",
         "MFMA spans of the layers above", "barrier between the spans and the cell updates", "cell updates + granule stores",
         "mask multipliers of the next phase", "publish -> every awaited granule seen", "values into LDS (masks applied)", "head",
         "end-of-phase barrier"]
for role, v in (("wave 0", vv[:16]),):
    mhz = v[10] / v[11] * 100
    print(f"{name} n={n} T={T}: kernel (cluster 0, member 0, {role}) {v[10]:.0f} cycles = {v[11] / 100:.2f} us at {mhz:.0f} MHz; {P} phases")
    for k, nm in enumerate(names):
        if v[k] == 0:
            continue
        per = f"{v[k] / P:7.0f} cycles = {v[k] / P / mhz:5.2f} us per phase" if k not in (0, 8) else ""
        print(f"  {nm:72s} {v[k]:8.0f} cycles {v[k] / mhz:6.2f} us  {v[k] / v[10] * 100:5.1f} %  {per}")
m.check()
