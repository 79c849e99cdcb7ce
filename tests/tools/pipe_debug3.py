import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd.estimate import nn_models
N = 16384
I, H, O = 22, 256, 14
x = torch.randn(N, I, device="cuda")
xn = x.cpu().numpy()
eye = np.eye(H, dtype=np.float32)
lrelu = lambda v: np.where(v > 0, v, np.float32(0.01) * v)
np.set_printoptions(precision=4, suppress=True, linewidth=250)
def full_h(mode):
    sd = orc.make_ff_state_dict(I, H, 2, O, 5)
    sd["_hidden_layers.0.weight"] = eye.copy(); sd["_hidden_layers.1.weight"] = eye.copy()
    for k in ["_hidden_layers.0.bias", "_hidden_layers.1.bias", "_output_layer.bias"]: sd[k] = np.zeros_like(sd[k])
    if mode == "b0=0": sd["_input_layer.bias"][:] = 0
    if mode == "W0=0": sd["_input_layer.weight"][:] = 0
    if mode == "b0=unit": sd["_input_layer.weight"][:] = 0; sd["_input_layer.bias"][:] = np.arange(H) + 1
    got = np.zeros((N, H), np.float32)
    m = nn_models.DropoutFF(O, H, 2, I, dropout=0.2, device=0)
    for base in range(0, H, O):
        pick = np.arange(base, min(base + O, H))
        w = np.zeros((O, H), np.float32); w[np.arange(len(pick)), pick] = 1.0
        sd["_output_layer.weight"] = w
        m.load_state_dict(sd)
        y = m(x).cpu().numpy(); m.check()
        got[:, pick] = y[:, :len(pick)]
    exp = lrelu(lrelu(lrelu(xn @ sd["_input_layer.weight"].T + sd["_input_layer.bias"])))
    return got, exp
for mode in ["b0=unit", "b0=0"]:
    got, exp = full_h(mode)
    bad = np.abs(got - exp) > 1e-5 * np.maximum(1, np.abs(exp))
    print(mode, "bad fraction", bad.mean())
    print(" bad per unit%64", bad.reshape(N, 4, 64).mean(axis=(0, 1)))
    print(" bad per wave", bad.reshape(N, 4, 64).mean(axis=(0, 2)))
    print(" bad per row%32", bad.reshape(-1, 32, H).mean(axis=(0, 2)))
    if mode == "b0=unit":
        for r in [0, 1, 32]:
            print(" row", r, "got", got[r].astype(int).tolist())
