"""MC-dropout forward latency (monte_carlo_predictions: one window, n rows, Philox inter-layer dropout)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd.estimate import nn_models
cfg = orc.MODEL_CONFIGS["pocket"]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0); m.load_state_dict(sd)
x = torch.randn(1, 6, cfg["I"], device="cuda")
for n in (1, 4, 16, 17, 25, 32, 33, 60, 64, 100, 256):
    for _ in range(20): m.monte_carlo_predictions(n, x, last_step_only=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): y = m.monte_carlo_predictions(n, x, last_step_only=True)
    b.record(); b.synchronize()
    m.check()
    print(f"mc={n:4d}: {a.elapsed_time(b) * 10:.1f} us per call  (spread over rows {float(y.std(dim=0).mean()):.4f})")
