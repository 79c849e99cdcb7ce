import sys, os
os.environ["APE_HIP_LIB"] = "/root/repo/arm-pose-estimation_amd/lib/diag/libape_hip_c16" + sys.argv[1] + ".so" if sys.argv[1] != "prod" else "/root/repo/arm-pose-estimation_amd/lib/libape_hip.so"
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd.estimate import nn_models
cfg = orc.MODEL_CONFIGS["uarm"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 3))
np.set_printoptions(precision=1, linewidth=250)
torch.manual_seed(0)
for B, T in [(288, 1), (288, 2), (288, 3), (1024, 64)]:
    x = torch.randn(B, T, cfg["I"], device="cuda")
    y2 = m.set_kernel("cluster")(x, last_step_only=True).cpu().numpy()[:, 0]
    m.check()
    y0 = m.set_kernel("tile16")(x, last_step_only=True).cpu().numpy()[:, 0]
    d = np.abs(y2 - y0).max(axis=1)
    print(sys.argv[1], B, T, "max", d.max(), "rows 0-15:", d[:16].max(), "rows 16-31:", d[16:32].max(), "bad rows", int((d > 1e-5).sum()), np.nonzero(d > 1e-5)[0][:40])
