"""What each part of the second-generation fp16 kernel really costs: launch time with the part switched off (diagnostic
library only -- `make -C csrc diag`; outputs of the ablated runs are wrong by construction).
APE_HIP_LIB=arm-pose-estimation_amd/lib/diag/libape_hip_diag.so python tests/tools/ablate_f16v2.py"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models

B, T = 1024, 64
cfg = orc.MODEL_CONFIGS["watch"]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0); m.load_state_dict(sd)
m.set_precision("f16")
x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
lib = _hip.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
NOEX, NOACT, NOMFMA, NOX, WT = 0x40000000, 0x20000000, 0x04000000, 0x00200000, 0x08000000

def run(n, flags):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, flags, None, 0.0, 0, C.c_void_p(y.data_ptr()), st), "fwd")
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3

base = None
for tag, fl in (("everything on", 0), ("no exchange (flags, gather, commit, publish)", NOEX), ("no gate transcendentals", NOACT),
                ("no MFMAs", NOMFMA), ("no x staging", NOX), ("no exchange, no x staging", NOEX | NOX),
                ("no exchange, no gates", NOEX | NOACT), ("no exchange, no MFMA", NOEX | NOMFMA),
                ("no exchange, gates, MFMA, x (skeleton)", NOEX | NOACT | NOMFMA | NOX), ("write-through exchange", WT)):
    run(30, fl)
    v = np.median([run(20, fl) for _ in range(7)])
    base = base or v
    print(f"{tag:48s} {v:8.1f} us   {v - base:+8.1f}   {(v / (2 * (T + 2))) * 2.39e3:7.0f} cycles per section", flush=True)
m.check()
