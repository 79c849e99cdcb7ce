"""A stream bank of S streams stepped for some frames (for rocprofv3 --kernel-trace --stats): python tests/tools/bank_trace.py [S] [n_mc] [frames] [auto|auto_gen1]"""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 100
kern = sys.argv[4] if len(sys.argv) > 4 else "auto"
cfg = orc.MODEL_CONFIGS["pocket"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
m.set_norm_stats(np.zeros(22), np.ones(22), np.zeros(14), np.ones(14))
m.set_body(orc.DEFAULT_BODY)
m.set_kernel(kern)
rng = np.random.default_rng(3)
rows = [torch.from_numpy(rng.normal(size=(S, 55)).astype(np.float32)).cuda() for _ in range(4)]
bank = StreamBank(m, S, 6, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=(n_mc or None), dropout=0.2)
for f in range(6):
    bank.push_rows(rows[f % 4], _hip.PARSE_WATCH_PHONE_POCKET); bank.step_datagrams()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for f in range(frames):
    bank.push_rows(rows[f % 4], _hip.PARSE_WATCH_PHONE_POCKET); bank.step_datagrams()
b.record(); b.synchronize()
if not (len(sys.argv) > 5 and sys.argv[5] == 'nocheck'): m.check()
print(f"S={S} n_mc={n_mc or 1} kernel={kern}: {a.elapsed_time(b) / frames * 1e3:.1f} us per frame of all streams")
