import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
cfg = orc.MODEL_CONFIGS["pocket"]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0); m.load_state_dict(sd)
lib = _hip.lib()
sig = lambda v: 1 / (1 + np.exp(-v))
def cell(W, U, b, x, h, c, keep=None):
    z = W @ x + U @ h + b
    H = 256
    i, f, g, o = sig(z[:H]), sig(z[H:2*H]), np.tanh(z[2*H:3*H]), sig(z[3*H:])
    c2 = f * c + i * g
    return o * np.tanh(c2), c2
B, T = 1, 1
x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
xn = x.cpu().numpy()[0, 0].astype(np.float64)
z = np.zeros(256)
b0 = sd['lstm.bias_ih_l0'] + sd['lstm.bias_hh_l0']; b1 = sd['lstm.bias_ih_l1'] + sd['lstm.bias_hh_l1']
h0, _ = cell(sd['lstm.weight_ih_l0'], sd['lstm.weight_hh_l0'], b0, xn, z, z)
def head(h): return sd['output_layer.weight'] @ h + sd['output_layer.bias']
hyp = {}
h1, _ = cell(sd['lstm.weight_ih_l1'], sd['lstm.weight_hh_l1'], b1, h0, z, z); hyp["correct"] = head(h1)
h1z, _ = cell(sd['lstm.weight_ih_l1'], sd['lstm.weight_hh_l1'], b1, z, z, z); hyp["layer1 saw zeros"] = head(h1z)
hyp["head saw zeros"] = head(z)
hyp["head saw h0"] = head(h0)
for name, sel in (("even units only", slice(0, None, 2)), ("odd units only", slice(1, None, 2)), ("first half", slice(0, 128)), ("second half", slice(128, 256))):
    hm = np.zeros(256); hm[sel] = h0[sel]
    hh, _ = cell(sd['lstm.weight_ih_l1'], sd['lstm.weight_hh_l1'], b1, hm, z, z); hyp["layer1 saw h0 " + name] = head(hh)
    hm = np.zeros(256); hm[sel] = h1[sel]; hyp["head saw h1 " + name] = head(hm)
_hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, 0, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
torch.cuda.synchronize()
yy = y.cpu().numpy()[0]
for k, v in hyp.items():
    print(f"{k:40s} {float(np.abs(yy - v).max()):.2e}")
m.check()
