"""all-steps forward (DropoutLSTM.forward semantics, [B,T,O] out): cluster kernel + head rows vs the batch-tile kernel"""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd.estimate import nn_models
cfg = orc.MODEL_CONFIGS["pocket"]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0); m.load_state_dict(sd)
x = torch.randn(1024, 64, cfg["I"], device="cuda")
for kern in ("auto", "tile16"):
    m.set_kernel(kern)
    for _ in range(30): m(x)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): y = m(x)
    b.record(); b.synchronize(); m.check()
    print(f"all steps, 1024 x 64, kernel {kern}: {a.elapsed_time(b) / 30 * 1e3:.0f} us per call")
