"""Where a section of the upper-layer kernel of the Monte-Carlo bank goes (diagnostic library, `make -C csrc diag`): blocking tops
and shader-clock sums of cluster 0 / member 0 / wave 0 over the last launch.
APE_HIP_LIB=arm-pose-estimation_amd/lib/diag/libape_hip_diag.so python tests/tools/diag_upper32.py [S] [n_mc]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = orc.MODEL_CONFIGS["pocket"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
m.set_norm_stats(np.zeros(22), np.ones(22), np.zeros(14), np.ones(14)); m.set_body(orc.DEFAULT_BODY)
rows = torch.randn(S, 55, device="cuda")
bank = StreamBank(m, S, 6, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc, dropout=0.2)
for f in range(20):
    bank.push_rows(rows, _hip.PARSE_WATCH_PHONE_POCKET); bank.step_datagrams()
torch.cuda.synchronize(); m.check()
lib = _hip.lib()
lib.ape_debug_read_wg.restype, lib.ape_debug_read_wg.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
buf = (C.c_ulonglong * (256 * 8))()
assert lib.ape_debug_read_wg(m.handle, buf) == 0
bx, bh, n, top, chain, tail = list(buf[16:22])
print(f"S={S} n_mc={n_mc}: {n} sections of cluster 0; blocking tops: {bx} for the input tile, {bh} for the gathered slices")
print(f"cycles per section: top {top / n:.0f}  MFMA chain {chain / n:.0f}  gates + publish {tail / n:.0f}  (sum {(top + chain + tail) / n:.0f}; "
      f"MFMAs alone: {(n - n // 6) * 16384 / n + (n // 6) * 8192 / n:.0f})")
