"""Soak run of the LSTM entry point: random shapes, kernels and dropout modes back to back for a fixed time, every result
checked against the batch-tile kernel on the same inputs and the cluster health word after every call.
python tests/tools/soak.py [seconds]"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
import os as _os; sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _load; _load.start()      # APE_SOAK_LOAD=1: beside device copies on a second stream
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(12345)
lib = _hip.lib()
models = {}
for name in ("pocket", "uarm", "watch"):
    cfg = orc.MODEL_CONFIGS[name]
    sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5)
    m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0); m.load_state_dict(sd)
    models[name] = (m, cfg)
ff = nn_models.DropoutFF(14, 256, 2, 22, dropout=0.2, device=0)
ff.load_state_dict(orc.make_ff_state_dict(22, 256, 2, 14, 6))
ff.set_norm_stats(0.1 * np.arange(22) - 1.0, 0.5 + 0.05 * np.arange(22), np.zeros(14), np.ones(14))
t0 = time.time(); n = 0; worst = 0.0; kinds = {}
t_say = t0
while time.time() - t0 < budget:
    if time.time() - t_say > 60:          # (a silent run is taken to be hung)
        t_say = time.time()
        print(f"  ... {n} calls in {t_say - t0:.0f} s", flush=True)
    if rng.integers(6) == 0:          # the MLP regressor: the two-stage pipeline (AUTO, N >= 64 rows per CU) against the tile kernel
        N = int(rng.choice([1, 100, 5000, 16383, 16384, 16385, 20000, 40001, 65536, 100000]))
        norm = bool(rng.integers(2))
        x = torch.randn(N, 22, device="cuda")
        ya = ff.set_kernel("auto")(x, normalize_input=norm)
        yb = ff.set_kernel("tile16")(x, normalize_input=norm)
        torch.cuda.synchronize()
        ff.check()
        err = float((ya - yb).abs().max())
        assert np.isfinite(err) and err < 2e-6, ("ff", N, norm, err)
        worst = max(worst, err)
        n += 1
        kinds[("ff", "f32", norm)] = kinds.get(("ff", "f32", norm), 0) + 1
        continue
    name = ("pocket", "uarm", "watch")[rng.integers(3)]
    m, cfg = models[name]
    B = int(rng.choice([1, 1, 1, 2, 2, 3, 4, 4, 5, 16, 17, 25, 60, 64, 100, 333, 512, 1000, 1024, 1025, 1500, 2049, 4096, 4500]))
    T = int(rng.choice([1, 2, 5, 6, 8, 9, 20, 64]))
    philox = bool(rng.integers(2))
    prec = "f16" if (not philox and rng.integers(5) == 0) else "f32"
    x = torch.randn(B, T, cfg["I"], device="cuda")
    flags = _hip.FLAG_DROPOUT_PHILOX if philox else 0
    ys = {}
    for kern in ("auto", "tile16"):
        m.set_kernel(kern); m.set_precision(prec if kern == "auto" else "f32")
        y = torch.full((B, cfg["O"]), float("nan"), device="cuda")
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, flags, None, 0.2 if philox else 0.0, n,
                                        C.c_void_p(y.data_ptr()), None), "fwd")
        ys[kern] = y
    torch.cuda.synchronize()
    m.check()
    a, b = ys["auto"].cpu().numpy(), ys["tile16"].cpu().numpy()
    assert np.isfinite(a).all(), (name, B, T, philox, prec)
    if not philox or B <= 512:       # (beyond one launch the cluster chunks re-key their masks: different samples)
        err = float(np.abs(a - b).max())
        tol = 5e-3 if prec == "f16" else 2e-5
        assert err < tol, (name, B, T, philox, prec, err)
        worst = max(worst, err if prec == "f32" else 0.0)
    n += 1
    kinds[(name, prec, philox)] = kinds.get((name, prec, philox), 0) + 1
m.set_kernel("auto"); m.set_precision("f32")
print(f"soak: {n} random calls in {time.time() - t0:.0f} s, all finite, health word clean, worst f32 |auto - tile16| = {worst:.1e}; mix: {len(kinds)} (model, precision, dropout) kinds")
