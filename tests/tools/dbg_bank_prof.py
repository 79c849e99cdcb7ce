import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
S, n_mc = int(sys.argv[1]), int(sys.argv[2])
cfg = orc.MODEL_CONFIGS["pocket"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
m.set_norm_stats(np.zeros(22), np.ones(22), np.zeros(14), np.ones(14)); m.set_body(orc.DEFAULT_BODY)
rows = [torch.randn(S, 55, device="cuda") for _ in range(4)]
bank = StreamBank(m, S, 6, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc, dropout=0.2)
for mode in ("plain", "profile", "plain"):
    bank.profile(mode == "profile")
    for f in range(8):
        bank.push_rows(rows[f % 4], _hip.PARSE_WATCH_PHONE_POCKET); bank.step_datagrams()
        try:
            m.check()
        except UserWarning as e:
            print(mode, "frame", f, "ABORT", str(e)[:80]); 
    if mode == "profile": print("profile_read", bank.profile_read())
print("done", m.stats())
