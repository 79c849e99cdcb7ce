"""Prefetch statistics of the second-generation f32 cluster kernel (diagnostic library, `make -C csrc diag`): how many gathers
of cluster 0 / member 0 / wave 0 were prefetched by the section in front, how many took the blocking form.
APE_HIP_LIB=arm-pose-estimation_amd/lib/diag/libape_hip_diag.so python tests/tools/diag_cluster32.py"""
import ctypes as C, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
B, T = 1024, 64
cfg = orc.MODEL_CONFIGS["pocket"]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0); m.load_state_dict(sd)
m.set_kernel("cluster")
x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
lib = _hip.lib()
lib.ape_debug_read_wg.restype, lib.ape_debug_read_wg.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
for _ in range(50):
    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, 0, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 8))()
assert lib.ape_debug_read_wg(m.handle, buf) == 0
v = list(buf[16:20])
print(f"last launch, T = {T}: gathers of layer 0's slices: {v[2]} prefetched, {v[0]} blocking; of layer 1's: {v[3]} prefetched, {v[1]} blocking")
m.check()
