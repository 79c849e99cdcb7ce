import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd.estimate import nn_models
N = 16384
I, H, O = 22, 256, 14
x = torch.randn(N, I, device="cuda")
sd = orc.make_ff_state_dict(I, H, 2, O, 5)
eye = np.eye(H, dtype=np.float32)
sd["_hidden_layers.0.weight"] = eye.copy(); sd["_hidden_layers.1.weight"] = eye.copy()
for k in ["_hidden_layers.0.bias", "_hidden_layers.1.bias", "_output_layer.bias"]: sd[k] = np.zeros_like(sd[k])
pick = [3, 40, 77, 100, 129, 166, 203, 255, 8, 64, 96, 160, 192, 224]
w = np.zeros((O, H), np.float32); w[np.arange(O), pick] = 1.0
sd["_output_layer.weight"] = w
m = nn_models.DropoutFF(O, H, 2, I, dropout=0.2, device=0)
m.load_state_dict(sd)
m.set_kernel("auto"); y1 = m(x).cpu().numpy(); m.check()
xn = x.cpu().numpy()
lrelu = lambda v: np.where(v > 0, v, np.float32(0.01) * v)
h = lrelu(lrelu(lrelu(xn @ sd["_input_layer.weight"].T + sd["_input_layer.bias"])))
exp = h[:, pick]
np.set_printoptions(precision=4, suppress=True, linewidth=220)
ok = np.abs(exp - y1) < 1e-5
print("fraction right per pick", ok.mean(axis=0))
print("fraction right per row%32", ok.reshape(-1, 32, O).mean(axis=(0, 2)))
print("fraction right per tile (first 16)", ok.reshape(-1, 32, O).mean(axis=(1, 2))[:16])
for r in [0, 1, 2, 33]:
    print(r, "exp", exp[r]); print(r, "got", y1[r])
