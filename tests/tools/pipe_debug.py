"""MLP pipeline kernel: structured weights to find which layer is off"""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd.estimate import nn_models
N = 16384
I, H, O = 22, 256, 14
x = torch.randn(N, I, device="cuda")
def run(tag, edit):
    sd = orc.make_ff_state_dict(I, H, 2, O, 5)
    edit(sd)
    m = nn_models.DropoutFF(O, H, 2, I, dropout=0.2, device=0)
    m.load_state_dict(sd)
    m.set_kernel("tile16"); y0 = m(x).cpu().numpy()
    m.set_kernel("auto"); y1 = m(x).cpu().numpy()
    m.check()
    d = np.abs(y0 - y1)
    print(f"{tag:28s} max diff {d.max():.3e}  ref scale {np.abs(y0).max():.3e}  cols off {np.nonzero(d.max(axis=0) > 1e-5)[0][:14]}", flush=True)
eye = np.eye(H, dtype=np.float32)
zb = lambda sd, ks: [sd.__setitem__(k, np.zeros_like(sd[k])) for k in ks]
def e_all_id(sd):
    sd["_hidden_layers.0.weight"] = eye.copy(); sd["_hidden_layers.1.weight"] = eye.copy()
    zb(sd, ["_hidden_layers.0.bias", "_hidden_layers.1.bias"])
def e_l1(sd):
    sd["_hidden_layers.1.weight"] = eye.copy(); zb(sd, ["_hidden_layers.1.bias"])
def e_l2(sd):
    sd["_hidden_layers.0.weight"] = eye.copy(); zb(sd, ["_hidden_layers.0.bias"])
def e_bias_only(sd):
    for k in ["_hidden_layers.0.weight", "_hidden_layers.1.weight"]: sd[k] = eye.copy()
def e_out_one(sd):
    e_all_id(sd)
    w = np.zeros((O, H), np.float32); w[np.arange(O), np.arange(O) * 17] = 1.0
    sd["_output_layer.weight"] = w; zb(sd, ["_output_layer.bias"])
run("identity hidden, no bias", e_all_id)
run("identity hidden, biases", e_bias_only)
run("layer 1 random, 2 identity", e_l2)
run("layer 2 random, 1 identity", e_l1)
run("identity hidden, out picks", e_out_one)
run("all random", lambda sd: None)
