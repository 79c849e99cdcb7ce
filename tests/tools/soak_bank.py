"""Soak run of the stream bank: random model / stream count / smoothing / Monte-Carlo settings, long frame sequences with
resets in between; every frame's messages and tails are recomputed from the oracle's window + smoothing bookkeeping
(WindowOracle) with the sample rows taken from ape_lstm_forward on explicitly built windows, and the oracle's FK /
message code.  python tests/tools/soak_bank.py [seconds]"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(777)
lib = _hip.lib()
body = orc.DEFAULT_BODY
t0 = time.time(); n_frames = 0; n_banks = 0; worst = 0.0
while time.time() - t0 < budget:
    name = ("pocket", "uarm", "watch")[rng.integers(3)]
    cfg = orc.MODEL_CONFIGS[name]
    sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], int(rng.integers(100)))
    m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0); m.load_state_dict(sd)
    m.set_body(body)
    S = int(rng.choice([1, 2, 5, 17, 64]))
    smooth = int(rng.choice([1, 2, 5]))
    n_mc = int(rng.choice([0, 0, 3, 25]))
    if rng.integers(6) == 0: S, n_mc, smooth = 350, 25, 1          # the shared-layer-0 route (8750 sample rows)
    T, I, O = cfg["T"], cfg["I"], cfg["O"]
    seed = int(rng.integers(1 << 40))
    bank = StreamBank(m, S, T, smooth=smooth, normalize=False, dtype=torch.float64,
                      monte_carlo_samples=(n_mc or None), dropout=0.2, seed=seed)
    k = max(1, n_mc)
    calls = 0
    for rnd in range(2):
        F = int(rng.integers(1, 2 * T + 3))
        shadow = [orc.WindowOracle(T, 1, None, lambda h: np.zeros((1, O))) for _ in range(S)]
        samples = {}
        wins = [orc.WindowOracle(T, smooth, None, (lambda h, s=s: samples[s])) for s in range(S)]
        for f in range(F):
            xx = rng.normal(size=(S, I)).astype(np.float32)
            bank.push_features(torch.from_numpy(xx).cuda())
            msg, tail = bank.step(with_tail=True)
            msg, tail = msg.cpu().numpy(), tail.cpu().numpy()
            hist = []
            for s in range(S):
                shadow[s].push(xx[s]); hist.append(np.vstack(shadow[s].rows).astype(np.float32))
            x = torch.from_numpy(np.repeat(np.stack(hist), k, axis=0)).cuda()
            y = torch.empty((S * k, O), dtype=torch.float32, device="cuda")
            drop = n_mc > 0 and cfg["L"] > 1
            if S * k >= 8192: m.set_kernel("tile16")                 # the shared route draws the batch-tile kernel's masks
            _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), S * k, T, _hip.FLAG_DROPOUT_PHILOX if drop else 0,
                                            None, 0.2 if drop else 0.0, seed + calls, C.c_void_p(y.data_ptr()), None), "fwd")
            torch.cuda.synchronize(); m.set_kernel("auto")
            calls += 1
            y = y.cpu().numpy().astype(np.float64)
            check = range(S) if S <= 64 else list(range(0, S, 50))
            for s in range(S):
                samples[s] = y[s * k:(s + 1) * k]
                pred = wins[s].push(xx[s])
                if s in check:
                    est = orc.arm_pose_from_targets(pred, body, cfg["layout"], "closed")
                    ref = orc.msg_from_est(est, body, cfg["layout"])
                    worst = max(worst, float(np.abs(msg[s] - ref).max()), float(np.abs(tail[s] - est[:, :6]).max()))
            n_frames += 1
        bank.reset()
    m.check()
    assert worst < 5e-5, (name, S, smooth, n_mc, worst)
    n_banks += 1
    del bank, m
print(f"bank soak: {n_banks} random banks, {n_frames} frames in {time.time() - t0:.0f} s, worst |msg/tail - recomputation| = {worst:.1e}")
