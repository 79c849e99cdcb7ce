"""First frames of a FRESH process on a Monte-Carlo bank's weight-stationary route against the batch-tile route (same Philox rows):
python tests/tools/cold_bank.py <pocket|uarm|watch> <S> <n_mc>   -> one line, 'OFF' when a row differs beyond the budget.
On the test-hooks library (APE_HIP_LIB = lib/diag/libape_hip_testhooks.so) the cold frame runs under INJECTED masks first and is compared
with the oracle's masked cell loop row by row (1e-6): the check then does not lean on the batch-tile route being right on a cold chip.
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
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5)
feats = np.random.default_rng(1).normal(size=(3, S, cfg["I"])).astype(np.float32)
res, kern = {}, {}
import ctypes as C
from wear_mocap_ape_amd import _hip
lib = _hip.lib()
orc_note = ""
if hasattr(lib, "ape_debug_set_bank_masks"):           # the process's FIRST launch of the route, under injected masks, against the oracle
    lib.ape_debug_set_bank_masks.restype, lib.ape_debug_set_bank_masks.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    lib.ape_debug_bank_targets.restype, lib.ape_debug_bank_targets.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    rows_n, H, L, O = S * n_mc, cfg["H"], cfg["L"], cfg["O"]
    rng = np.random.default_rng(7)
    masks = [(rng.random((rows_n, T, H)) >= 0.2).astype(np.float32) / np.float32(0.8) for _ in range(L - 1)]
    md = torch.from_numpy(np.stack(masks)).cuda()
    bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=11)
    assert lib.ape_debug_set_bank_masks(bank._handle, C.c_void_p(md.data_ptr())) == 0
    bank.push_features(torch.from_numpy(feats[0]).cuda())
    bank.step()
    y = np.empty((rows_n, O), dtype=np.float32)
    assert lib.ape_debug_bank_targets(bank._handle, y.ctypes.data_as(C.c_void_p)) == 0
    win = np.repeat(feats[0][:, None, :], T, axis=1)           # a cold window: the new row fills every step (estimator.py:96-97)
    ref = orc.lstm_forward(sd, np.repeat(win, n_mc, axis=0), masks=masks)[:, -1, :]
    d_o = float(np.abs(y - ref).max())
    orc_note = f"; cold frame under injected masks vs oracle {d_o:.2e} [{m.last_kernel()}]" + ("  OFF (oracle)" if d_o > 1e-6 else "")
    assert lib.ape_debug_set_bank_masks(bank._handle, None) == 0
    m.check()
    del bank
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
print(f"{name} bank {S} x {n_mc} {kern['auto']} vs {kern['tile16']}: cold frame {d[0]:.2e}, then {d[1]:.2e} {d[2]:.2e}" + (f"  OFF ({rows} rows)" if max(d) > 2e-5 else "") + orc_note)
