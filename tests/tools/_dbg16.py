import sys, os, ctypes as C
os.environ["APE_HIP_LIB"] = "/root/repo/arm-pose-estimation_amd/lib/diag/libape_hip_c16dump.so"
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
cfg = orc.MODEL_CONFIGS["uarm"]
I, H, L, O = cfg["I"], cfg["H"], cfg["L"], cfg["O"]
sd = orc.make_state_dict(I, H, L, O, 3)
m = nn_models.DropoutLSTM(I, H, L, O, dropout=0.2, device=0)
m.load_state_dict(sd)
np.set_printoptions(precision=2, linewidth=250)
B, T = 288, int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn(B, T, I, device="cuda")
y2 = m.set_kernel("cluster")(x, last_step_only=True).cpu().numpy()[:, 0]
m.check()
lib = _hip.lib()
buf = (C.c_ulonglong * 2048)()
lib.ape_debug_read_wg.argtypes = [C.c_void_p, C.c_void_p]
lib.ape_debug_read_wg(m.handle, buf)
raw = np.frombuffer(buf, dtype=np.float32).reshape(-1)
dump = raw[:3072].reshape(L, 2, 4, 2, 64)      # [l][t][wave][rt][lane]
xs = x[:32].cpu().numpy().astype(np.float64)
sig = lambda v: 1 / (1 + np.exp(-v))
inp = xs
for l in range(L):
    Wi, Wh = sd[f"lstm.weight_ih_l{l}"].astype(np.float64), sd[f"lstm.weight_hh_l{l}"].astype(np.float64)
    b = (sd[f"lstm.bias_ih_l{l}"] + sd[f"lstm.bias_hh_l{l}"]).astype(np.float64)
    h = np.zeros((32, H)); c = np.zeros((32, H)); outs = []
    for t in range(T):
        z = inp[:, t] @ Wi.T + h @ Wh.T + b
        i, f, g, o = sig(z[:, :H]), sig(z[:, H:2*H]), np.tanh(z[:, 2*H:3*H]), sig(z[:, 3*H:])
        c = f * c + i * g; h = o * np.tanh(c); outs.append(h)
        if t < 2:
            for w in range(4):
                for rt in range(2):
                    got = dump[l, t, w, rt]
                    exp = np.array([h[rt * 16 + (lane & 15), 4 * w + (lane >> 4)] for lane in range(64)])
                    print(f"layer {l} step {t} wave {w} rt {rt}: max |dump - numpy| {np.abs(got - exp).max():.2e}")
    inp = np.stack(outs, axis=1)
y0 = m.set_kernel("tile16")(x, last_step_only=True).cpu().numpy()[:, 0]
print("per row |c16 - tile16| (first 64):", np.abs(y2 - y0).max(axis=1)[:64])
