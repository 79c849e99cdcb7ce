"""Where a section of the 3 x 128 Monte-Carlo bank's upper-layer kernel goes (diagnostic library, `make -C csrc diag`): blocking tops and
shader-clock sums of cluster 0 / member 0 / wave 0 over the last launch.
APE_HIP_LIB=arm-pose-estimation_amd/lib/diag/libape_hip_diag.so python tests/tools/diag_upper128.py [S] [n_mc]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = orc.MODEL_CONFIGS["uarm"]
T = cfg["T"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
m.set_norm_stats(np.zeros(cfg["I"]), np.ones(cfg["I"]), np.zeros(cfg["O"]), np.ones(cfg["O"])); m.set_body(orc.DEFAULT_BODY)
rows = torch.randn(S, 55, device="cuda")
bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc, dropout=0.2)
for f in range(12):
    bank.push_rows(rows, _hip.PARSE_WATCH_PHONE_UARM); bank.step_datagrams()
torch.cuda.synchronize(); m.check()
assert m.last_kernel() == "ape_lstm_upper128", m.last_kernel()
lib = _hip.lib()
lib.ape_debug_read_wg.restype, lib.ape_debug_read_wg.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
buf = (C.c_ulonglong * (256 * 8))()
assert lib.ape_debug_read_wg(m.handle, buf) == 0
blk, n_go, n, top, chain, gates, pub = list(buf[16:23])
tiles = (S * n_mc + 31) // 32
bare = (2 * T - 2) * 4096 * 25 / n if False else None
print(f"S={S} n_mc={n_mc}: {n} sections of cluster 0 (one per step of a set, both layers), {blk} of them with a blocking top, {n_go} looks succeeded")
print(f"cycles per section: top {top / n:.0f}  MFMA chains {chain / n:.0f}  gates (+ head) {gates / n:.0f}  publish {pub / n:.0f}  "
      f"(sum {(top + chain + gates + pub) / n:.0f}); whole launch {(top + chain + gates + pub) / 1e6:.3f} M cycles, "
      f"MFMAs alone {(tiles / 64) * (2 * T - 1) * 2 * 4096 / 1e6:.3f} M")
k0 = np.array(buf[64:128], dtype=np.float64); k1 = np.array(buf[128:192], dtype=np.float64); k2 = np.array(buf[192:256], dtype=np.float64)
if k2.max() > 0:
    print(f"per cluster (member 0): entry -> first section {np.median(k1 - k0):.0f} cycles (max {(k1 - k0).max():.0f}); first section -> exit "
          f"median {np.median(k2 - k1) / 1e6:.3f} M, min {(k2 - k1).min() / 1e6:.3f} M, max {(k2 - k1).max() / 1e6:.3f} M; "
          f"classes (cluster % 8) medians {[round(float(np.median((k2 - k1)[c::8])) / 1e6, 3) for c in range(8)]}")
