"""A stream bank of S streams stepped for some frames (for rocprofv3 --kernel-trace --stats): python tests/tools/bank_trace.py [S] [n_mc] [frames] [auto|auto_gen1] [check|nocheck] [pocket|uarm|watch]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 100
kern = sys.argv[4] if len(sys.argv) > 4 else "auto"
name = sys.argv[6] if len(sys.argv) > 6 else "pocket"
cfg = orc.MODEL_CONFIGS[name]
T = cfg["T"]
kind = {"pocket": _hip.PARSE_WATCH_PHONE_POCKET, "uarm": _hip.PARSE_WATCH_PHONE_UARM, "watch": _hip.PARSE_WATCH_ONLY}[name]
width = _hip.PARSE_SHAPES[kind][0]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
m.set_norm_stats(np.zeros(cfg["I"]), np.ones(cfg["I"]), np.zeros(cfg["O"]), np.ones(cfg["O"]))
m.set_body(orc.DEFAULT_BODY)
m.set_kernel(kern)
rng = np.random.default_rng(3)
rows = [torch.from_numpy(rng.normal(size=(S, width)).astype(np.float32)).cuda() for _ in range(4)]
bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=(n_mc or None), dropout=0.2)
for f in range(T):
    bank.push_rows(rows[f % 4], kind); bank.step_datagrams()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for f in range(frames):
    bank.push_rows(rows[f % 4], kind); bank.step_datagrams()
b.record(); b.synchronize()
if not (len(sys.argv) > 5 and sys.argv[5] == 'nocheck'): m.check()
print(f"{name} S={S} n_mc={n_mc or 1} kernel={kern}: {a.elapsed_time(b) / frames * 1e3:.1f} us per frame of all streams")
