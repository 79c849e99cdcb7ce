"""Per-section cycle shares of the second-generation fp16 kernel (diagnostic library built by `make -C csrc diag`):
APE_HIP_LIB=arm-pose-estimation_amd/lib/diag/libape_hip_diag.so python tests/tools/stamps_f16v2.py [flags]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models

B, T = 1024, 64
flags = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
cfg = orc.MODEL_CONFIGS["watch"]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0); m.load_state_dict(sd)
m.set_precision("f16")
x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
lib = _hip.lib()
lib.ape_debug_read_wg.restype, lib.ape_debug_read_wg.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
for _ in range(30):
    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, flags, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 8))()
assert lib.ape_debug_read_wg(m.handle, buf) == 0
v = np.array(buf[:10], dtype=np.float64)
names = ["flag poll", "gather -> landed (+ other set's drain)", "flag raise + LDS commit + x staging", "barrier", "MFMA spans",
         "gates + own staging", "publish", "fragment reads issued + x staging"]
sections = 2 * (T + 2)
print(f"total {v[8]:.0f} cycles = {v[9] / 100:.1f} us ({v[8] / v[9] * 100:.0f} MHz), {sections} sections, {v[8] / sections:.0f} cycles per section (stamps cost extra)")
for k, n in enumerate(names):
    print(f"  {n:40s} {v[k]:10.0f} cycles  {v[k] / v[8] * 100:5.1f} %   {v[k] / sections:7.0f} per section")
print(f"  {'unaccounted (prologue, head, ...)':40s} {v[8] - v[:8].sum():10.0f}")
m.check()
