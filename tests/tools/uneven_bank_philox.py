"""A Monte-Carlo bank drawing its own (Philox) masks, frame by frame beside a queue of device copies, against ape_lstm_forward on the
batch-tile kernel with the same Philox key on explicitly repeated windows -- run IDLE (the copies drained first), and once more under load:
which side moves?  APE_HIP_LIB=.../lib/diag/libape_hip_testhooks.so python tests/tools/uneven_bank_philox.py [name] [S] [n_mc] [smooth] [frames] [copies]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
name = sys.argv[1] if len(sys.argv) > 1 else "watch"
S, n_mc, smooth, frames, ncopy = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 41), (3, 60), (4, 2), (5, 30), (6, 24)))
cfg = orc.MODEL_CONFIGS[name]
T, I, O = cfg["T"], cfg["I"], cfg["O"]
lib = _hip.lib()
lib.ape_debug_bank_targets.restype, lib.ape_debug_bank_targets.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
side = torch.cuda.Stream()
a = torch.full((64 << 20,), 1.0, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _load
DAEMON = _load.start()          # APE_SOAK_LOAD=1: a thread keeps copies in flight the whole time instead of the bursts below
def burst():
    if DAEMON: return
    with torch.cuda.stream(side):
        for _ in range(ncopy): b.copy_(a, non_blocking=True)
BANKS = int(os.environ.get('BANKS', '1'))        # fresh model + bank each time: first frames on untouched buffers
for bank_no in range(BANKS):
    sd = orc.make_state_dict(I, cfg["H"], cfg["L"], O, 7)
    m = nn_models.DropoutLSTM(I, cfg["H"], cfg["L"], O, dropout=0.2, device=0); m.load_state_dict(sd); m.set_body(orc.DEFAULT_BODY)
    print(f'-- bank {bank_no}', flush=True)
    seed = 123456789
    bank = StreamBank(m, S, T, smooth=smooth, normalize=False, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=seed)
    rng = np.random.default_rng(5)
    shadow = [orc.WindowOracle(T, 1, None, lambda h: np.zeros((1, O))) for _ in range(S)]
    rows = S * n_mc
    samples = {}
    stack = [orc.WindowOracle(T, smooth, None, (lambda h, s=s: samples[s])) for s in range(S)]      # the smoothing stack, fed the bank's OWN targets
    for f in range(frames):
        xx = rng.normal(size=(S, I)).astype(np.float32)
        bank.push_features(torch.from_numpy(xx).cuda())
        torch.cuda.synchronize()
        burst()
        msg_d, tail_d = bank.step(with_tail=True)
        kern = m.last_kernel()
        yb = np.empty((rows, O), dtype=np.float32)
        assert lib.ape_debug_bank_targets(bank._handle, yb.ctypes.data_as(C.c_void_p)) == 0
        side.synchronize(); torch.cuda.synchronize()
        hist = []
        for s in range(S):
            shadow[s].push(xx[s]); hist.append(np.vstack(shadow[s].rows).astype(np.float32))
        x = torch.from_numpy(np.repeat(np.stack(hist), n_mc, axis=0)).cuda()
        ys = []
        for loaded in (False, True):
            y = torch.empty((rows, O), dtype=torch.float32, device="cuda")
            m.set_kernel("tile16")
            torch.cuda.synchronize()
            if loaded: burst()
            _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), rows, T, _hip.FLAG_DROPOUT_PHILOX, None, 0.2, seed + f, C.c_void_p(y.data_ptr()), None), "fwd")
            torch.cuda.synchronize(); side.synchronize(); m.set_kernel("auto")
            ys.append(y.cpu().numpy())
        d_bank = np.abs(yb - ys[0]).max(axis=1); d_ref = np.abs(ys[1] - ys[0]).max(axis=1)
        bad_b, bad_r = np.nonzero(d_bank > 2e-6)[0], np.nonzero(d_ref > 0)[0]
        msg = f"frame {f} [{kern}]: bank under load vs idle batch-tile: {len(bad_b)} rows off (max {d_bank.max():.2e})"
        if len(bad_b): msg += f", streams {sorted(set(int(r) // n_mc for r in bad_b))[:12]}, tiles {sorted(set(int(r) // 32 for r in bad_b))[:12]}"
        msg += f"; batch-tile under load vs idle: {len(bad_r)} rows differ (max {d_ref.max():.2e})"
        # the post-filter alone: the bank's own targets through the oracle's stacking + FK against the tails the bank returned
        tail_d = tail_d.cpu().numpy(); worst_t, bad_s = 0.0, []
        for s in range(S):
            samples[s] = yb[s * n_mc:(s + 1) * n_mc].astype(np.float64)
            pred = stack[s].push(xx[s])
            est = orc.arm_pose_from_targets(pred, orc.DEFAULT_BODY, cfg["layout"], "closed")
            dt = float(np.abs(tail_d[s] - est[:, :6]).max())
            worst_t = max(worst_t, dt)
            if dt > 5e-6: bad_s.append(s)
        msg += f"; post-filter tails vs oracle on the bank's own targets: worst {worst_t:.2e}" + (f" streams {bad_s[:16]}" if bad_s else "")
        print(msg, flush=True)

    m.check()
    del bank, m
