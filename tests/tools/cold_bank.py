"""First frames of a FRESH process on a Monte-Carlo bank's weight-stationary route against the batch-tile route (same Philox rows):
python tests/tools/cold_bank.py <pocket|uarm|watch> <S> <n_mc>   -> one line, 'OFF' when a row differs beyond the budget.
(round 4: with plain hand-over stores the first launch of a fresh process read stale slices in 7 of 8 runs on lstm_upper128.hip)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
name, S, n_mc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = orc.MODEL_CONFIGS[name]
T = cfg["T"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
m.set_norm_stats(np.zeros(cfg["I"]), np.ones(cfg["I"]), np.zeros(cfg["O"]), np.ones(cfg["O"])); m.set_body(orc.DEFAULT_BODY)
feats = np.random.default_rng(1).normal(size=(3, S, cfg["I"])).astype(np.float32)
res, kern = {}, {}
for route in ("auto", "tile16"):                      # the cooperative route FIRST: its first launch is the process's cold one
    bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=11)
    out = []
    for f in range(3):
        m.set_kernel(route)
        bank.push_features(torch.from_numpy(feats[f]).cuda())
        msg, tail = bank.step(with_tail=True)
        out.append(tail.cpu().numpy().reshape(S * n_mc, 6).copy())
    kern[route] = m.last_kernel()
    m.set_kernel("auto"); m.check()
    res[route] = out
    del bank
d = [float(np.abs(res["auto"][f] - res["tile16"][f]).max()) for f in range(3)]
rows = int((np.abs(res["auto"][0] - res["tile16"][0]).max(axis=1) > 2e-5).sum())
print(f"{name} bank {S} x {n_mc} {kern['auto']} vs {kern['tile16']}: cold frame {d[0]:.2e}, then {d[1]:.2e} {d[2]:.2e}" + (f"  OFF ({rows} rows)" if max(d) > 2e-5 else ""))
