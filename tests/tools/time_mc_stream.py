"""One stream's frame step in the estimators' default mode (25 Monte-Carlo dropout samples, T = 6) through the device-side
stream bank, per-frame HIP events: python tests/tools/time_mc_stream.py [S] [n_mc]"""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 25
cfg = orc.MODEL_CONFIGS["pocket"]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0, target_layout=_hip.LAYOUT_ORI_CAL_LARM_UARM_HIPS)
m.load_state_dict(sd)
rng = np.random.default_rng(0)
I, O = cfg["I"], cfg["O"]
m.set_norm_stats(rng.normal(size=I), 1 + rng.random(I), rng.normal(size=O) * 0.1, 1 + 0.1 * rng.random(O))
rows = [torch.from_numpy(rng.normal(size=(S, 55)).astype(np.float32)).cuda() for _ in range(4)]
bank = StreamBank(m, S, 6, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc if n_mc > 1 else None, dropout=0.2)
us = []
for i in range(320):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    bank.push_rows(rows[i % 4], _hip.PARSE_WATCH_PHONE_POCKET)
    bank.step_datagrams()
    b.record(); b.synchronize()
    if i >= 20: us.append(a.elapsed_time(b) * 1e3)
m.check()
print(f"S={S} n_mc={n_mc}: frame step p50 {np.percentile(us, 50):.1f} us  p99 {np.percentile(us, 99):.1f} us")
