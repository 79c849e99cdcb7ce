"""debug helper: message kernel vs golden, per case"""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd.estimate import compose_msg
from wear_mocap_ape_amd.utility.names import NNS_TARGETS
np.set_printoptions(linewidth=200, precision=3)
for layout, tgt in ((0, NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS), (1, NNS_TARGETS.ORI_CAL_LARM_UARM)):
    g = np.load(f"/root/repo/tests/golden/fk_layout{layout}.npz")
    for tag in ("bd", "bo"):
        for N in (1, 7, 300):
            est = g[f"est_{tag}_N{N}"]; body = g[f"body_{tag}"]
            msg = compose_msg.msg_from_nn_targets_est(est, body, tgt)
            ref = g[f"msg_{tag}_N{N}"]
            print(layout, tag, N, "max", np.abs(msg - ref).max(), " orc", np.abs(orc.msg_from_est(est, body, layout) - ref).max())
            if np.abs(msg - ref).max() > 1e-9:
                print("  diff", np.abs(msg - ref))
                for n in (299, 256, 255, 200):
                    m2 = compose_msg.msg_from_nn_targets_est(est[:n], body, tgt)
                    print("   first", n, "rows vs oracle:", np.abs(m2 - orc.msg_from_est(est[:n], body, layout)).max())
