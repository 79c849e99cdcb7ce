"""Wall-clock timeline of ONE workgroup of lstm_cluster16.hip (diagnostic build, `make diag`; cluster 0 / member 0 / wave 0, the 100 MHz
s_memrealtime counter): prologue, every section, final publish / gather, head.  APE_C16_MIN_T=1 python tests/tools/timeline_uarm16.py [B] [T]"""
import ctypes as C, os, sys
os.environ.setdefault("APE_HIP_LIB", "/root/repo/arm-pose-estimation_amd/lib/diag/libape_hip_diag.so")
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = orc.MODEL_CONFIGS["uarm"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0))
m.set_kernel("cluster")
x = torch.randn(B, T, cfg["I"], device="cuda")
lib = _hip.lib(); buf = (C.c_ulonglong * 2048)()
lib.ape_debug_read_wg.argtypes = [C.c_void_p, C.c_void_p]
for rep in range(3):
    for _ in range(20): m(x, last_step_only=True)
    torch.cuda.synchronize(); m.check()
    lib.ape_debug_read_wg(m.handle, buf)
    d = np.frombuffer(buf, dtype=np.uint64)
    n = int(d[128]); t = d[129:129 + n].astype(np.float64) * 0.01      # microseconds
    names = ["entry", "prologue"] + [f"S({ph},{l})" for ph in range(T + 2) for l in range(3)] + ["final publish", "store drained+flag", "flags seen", "dma landed", "barrier", "head"]
    print(f"{m.last_kernel()} B={B} T={T}: total {t[-1] - t[0]:.2f} us")
    print("  " + "  ".join(f"{names[i] if i < len(names) else i}:{t[i] - t[i - 1]:.2f}" for i in range(1, n)))
    fl = d[256:320].astype(np.float64).reshape(8, 8) * 0.01 - t[0]
    en = d[320:328].astype(np.float64) * 0.01 - t[0]
    print("  entry of members 0..7 (us after member 0):", " ".join(f"{v:.2f}" for v in en))
    print("  last flag raised, member x wave (us after member 0's entry):")
    for mm in range(8): print("    m%d: " % mm + " ".join(f"{v:.2f}" for v in fl[mm]))
    print(f"  (member 0 wave 0: flags seen at {t[names.index('flags seen')] - t[0]:.2f})")
