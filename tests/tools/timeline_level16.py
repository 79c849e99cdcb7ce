"""Wall-clock timeline of ONE workgroup of lstm_level16.hip (diagnostic build, `make diag`; cluster 0 / member 0 / thread 0, the 100 MHz
s_memrealtime counter): prologue, every level (compute + hand-over), head.  python tests/tools/timeline_level16.py [B] [T]"""
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
x = torch.randn(B, T, cfg["I"], device="cuda")
lib = _hip.lib(); buf = (C.c_ulonglong * 2048)()
lib.ape_debug_read_wg.argtypes = [C.c_void_p, C.c_void_p]
for rep in range(3):
    for _ in range(20): m(x, last_step_only=True)
    torch.cuda.synchronize(); m.check()
    lib.ape_debug_read_wg(m.handle, buf)
    d = np.frombuffer(buf, dtype=np.uint64)
    for ag in (0, 1):
        base = 128 - 64 * ag
        n = min(int(d[base]), 63); t = d[base + 1:base + 1 + n].astype(np.float64) * 0.01      # microseconds
        print(f"{m.last_kernel()} B={B} T={T} agent {ag}: total {t[-1] - t[0]:.2f} us")
        dt = [t[i] - t[i - 1] for i in range(1, n)]
        print(f"  prologue {dt[0]:.2f}")
        for k in range((n - 3) // 5):
            w, c, xs, sw, ab = dt[1 + 5 * k: 6 + 5 * k]
            print(f"  level {k}: wait for the SIMD {w:.2f}  compute {c:.2f}  x staging {xs:.2f}  sweeps {sw:.2f}  agent barrier {ab:.2f}   = {w + c + xs + sw + ab:.2f}")
        print(f"  head {dt[-1]:.2f}")
        sb = d[1840 + 8 * ag:1840 + 8 * ag + 6].astype(np.float64) * 0.01
        print(f"  level 4 sweep: poll8 {sb[0] - sb[5]:.2f}  judge {sb[1] - sb[0]:.2f}  commit {sb[2] - sb[1]:.2f}  poll4 {sb[3] - sb[2]:.2f}  judge+commit {sb[4] - sb[3]:.2f}")
        print("  failed polls per level:", " ".join(str(int(v)) for v in d[1800 + 16 * ag:1800 + 16 * ag + T + 2]))
    e = d[256:256 + 3 * 512].astype(np.float64).reshape(512, 3)
    if B >= 1024:
        t = d[129:130].astype(np.float64) * 0.01
        t0 = e[:, 0].min() * 0.01
        ent, ext = e[:, 0] * 0.01 - t0, e[:, 1] * 0.01 - t0
        print(f"  all 512 workgroups: entry min/median/max {ent.min():.2f} / {np.median(ent):.2f} / {ent.max():.2f} us, exit min/median/max {ext.min():.2f} / {np.median(ext):.2f} / {ext.max():.2f} us")
        print("  entry by block index, every 32nd:", " ".join(f"{v:.1f}" for v in ent[::32]))
        print("  exit  by block index, every 32nd:", " ".join(f"{v:.1f}" for v in ext[::32]))
