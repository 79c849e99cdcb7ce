"""Where does a wrong bank frame start?  Fresh Monte-Carlo banks (Philox masks, the weight-stationary route of the 2 x 256 models), a few
frames each with NO host synchronisation between push and step (the soak's call pattern), beside device copies when APE_SOAK_LOAD=1.  Each
frame's NN targets are compared with the batch-tile kernel on explicitly built windows; on a difference the bank's own buffers are read back
(test-hooks library: ape_debug_bank_buffer) and compared with the host's windows (layer 0's input tiles) and a float64 layer 0 in numpy (its
output sequence): the first buffer / step / k-block / row that is off names the kernel boundary.
    APE_HIP_LIB=.../libape_hip_testhooks.so [APE_SOAK_LOAD=1] python tests/tools/locate_bank.py [seconds] [name] [S] [smooth] [n_mc] [frames]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _load; _load.start()
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
name = sys.argv[2] if len(sys.argv) > 2 else "watch"
S, smooth, n_mc, frames = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((3, 130), (4, 2), (5, 20), (6, 4)))
# (banks of up to 96 streams run launch A on the first-generation kernel's one-layer form since the end of round 5: their layer-0 sequence
#  is row-major in the model's workspace, not in the buffer read back below -- the read-back part of this tool is for S > 96)
SYNC_PUSH = os.environ.get("LOCATE_SYNC_PUSH") == "1"          # (experiment: a host synchronisation between push and step)
cfg = orc.MODEL_CONFIGS[name]
T, I, O, H = cfg["T"], cfg["I"], cfg["O"], cfg["H"]
lib = _hip.lib()
lib.ape_debug_bank_targets.restype, lib.ape_debug_bank_targets.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
lib.ape_debug_bank_buffer.restype = C.c_int
lib.ape_debug_bank_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
tiles = (S + 31) // 32


def fetch(bank, which, n_floats):
    out = np.empty(n_floats, dtype=np.float32); got = C.c_size_t(0)
    assert lib.ape_debug_bank_buffer(bank._handle, which, out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(got)) == 0
    assert got.value == out.nbytes, (which, got.value, out.nbytes)
    return out


def layer0(sd, x):
    """float64 layer 0 over x [S,T,I] -> [S,T,H]"""
    w_ih, w_hh = sd["lstm.weight_ih_l0"].astype(np.float64), sd["lstm.weight_hh_l0"].astype(np.float64)
    b = sd["lstm.bias_ih_l0"].astype(np.float64) + sd["lstm.bias_hh_l0"].astype(np.float64)
    h = np.zeros((x.shape[0], H)); c = np.zeros_like(h); out = np.empty((x.shape[0], T, H))
    sg = lambda v: 1.0 / (1.0 + np.exp(-v))
    for t in range(T):
        pre = x[:, t].astype(np.float64) @ w_ih.T + h @ w_hh.T + b
        c = sg(pre[:, H:2 * H]) * c + sg(pre[:, :H]) * np.tanh(pre[:, 2 * H:3 * H])
        h = sg(pre[:, 3 * H:]) * np.tanh(c)
        out[:, t] = h
    return out


rng = np.random.default_rng(777)
t0 = time.time(); n_banks = n_bad = n_frames = 0
prev_bank_hf = None          # the output sequence the bank in front left behind (the allocator hands the same addresses to the next one)


def whence(vec, pools):
    """where else do these 8 floats live?  nearest [.., 8] cell of every pool"""
    out = []
    for label, pool in pools:
        if pool is None: continue
        flat = pool.reshape(-1, 8).astype(np.float64)
        d = np.abs(flat - vec.astype(np.float64)).max(axis=1)
        i = int(d.argmin())
        out.append(f"{label}: nearest cell {np.unravel_index(i, pool.shape[:-1])} off by {d[i]:.2e}")
    return "; ".join(out)
while time.time() - t0 < budget:
    sd = orc.make_state_dict(I, H, cfg["L"], O, int(rng.integers(100)))
    m = nn_models.DropoutLSTM(I, H, cfg["L"], O, dropout=0.2, device=0); m.load_state_dict(sd); m.set_body(orc.DEFAULT_BODY)
    seed = int(rng.integers(1 << 40))
    bank = StreamBank(m, S, T, smooth=smooth, normalize=False, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=seed)
    shadow = [orc.WindowOracle(T, 1, None, lambda h: np.zeros((1, O))) for _ in range(S)]
    rows = S * n_mc
    prev_h0 = None
    for f in range(frames):
        xx = rng.normal(size=(S, I)).astype(np.float32)
        bank.push_features(torch.from_numpy(xx).cuda())
        if SYNC_PUSH: torch.cuda.synchronize()
        bank.step(with_tail=True)
        kern = m.last_kernel()
        yb = np.empty((rows, O), dtype=np.float32)
        assert lib.ape_debug_bank_targets(bank._handle, yb.ctypes.data_as(C.c_void_p)) == 0
        hist = []
        for s in range(S):
            shadow[s].push(xx[s]); hist.append(np.vstack(shadow[s].rows).astype(np.float32))
        hist = np.stack(hist)
        x = torch.from_numpy(np.repeat(hist, n_mc, axis=0)).cuda()
        y = torch.empty((rows, O), dtype=torch.float32, device="cuda")
        m.set_kernel("tile16"); torch.cuda.synchronize()
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), rows, T, _hip.FLAG_DROPOUT_PHILOX, None, 0.2, seed + f, C.c_void_p(y.data_ptr()), None), "fwd")
        torch.cuda.synchronize(); m.set_kernel("auto")
        dd = np.abs(yb - y.cpu().numpy()).max(axis=1)
        n_frames += 1
        if dd.max() <= 5e-6:
            if f == frames - 1: prev_bank_hf = fetch(bank, 2, tiles * T * 8192).reshape(tiles, T, 32, 32, 8)
            continue
        n_bad += 1
        print(f"bank {n_banks} frame {f} [{kern}]: {int((dd > 5e-6).sum())} rows off (max {dd.max():.2e}), streams {sorted(set(int(r) // n_mc for r in np.nonzero(dd > 5e-6)[0]))[:16]}", flush=True)
        # layer 0's input tiles [tile][t][k-block 4][row 32][8]
        xf = fetch(bank, 1, tiles * T * 1024).reshape(tiles, T, 4, 32, 8)
        want = np.zeros((tiles * 32, T, 32), dtype=np.float32); want[:S, :, :I] = hist
        want = want.reshape(tiles, 32, T, 4, 8).transpose(0, 2, 3, 1, 4)
        bad = np.argwhere(xf != want)
        print(f"    layer-0 input tiles: {len(bad)} floats differ" + (f"; (tile, t, k-block, row) {sorted(set(map(tuple, bad[:, :4].tolist())))[:12]}" if len(bad) else ""), flush=True)
        # layer 0's output sequence [tile][t][k-block 32][row 32][8] against float64 numpy
        hf = fetch(bank, 2, tiles * T * 8192).reshape(tiles, T, 32, 32, 8)
        h0 = np.zeros((tiles * 32, T, H)); h0[:S] = layer0(sd, hist)
        h0 = h0.reshape(tiles, 32, T, 32, 8).transpose(0, 2, 3, 1, 4)
        live = np.zeros((tiles, 32), dtype=bool); live.reshape(-1)[:S] = True
        dh = np.abs(hf - h0) * live[:, None, None, :, None]
        badh = np.argwhere(dh > 2e-6)
        print(f"    layer-0 output sequence: {len(badh)} floats off by > 2e-6 (max {dh.max():.2e})", flush=True)
        if len(badh):
            by_t = {}
            for tl, t, kb, row, j in badh.tolist(): by_t.setdefault((tl, t), set()).add((kb, row))
            for (tl, t), cells in sorted(by_t.items())[:8]:
                kbs, rws = sorted({c[0] for c in cells}), sorted({c[1] for c in cells})
                print(f"      tile {tl} step {t}: {len(cells)} (k-block, row) cells; k-blocks {kbs[:40]}; rows {rws}; max {dh[tl, t].max():.2e}", flush=True)
            tl, t, kb, row, _ = badh[0].tolist()
            h_prev = np.zeros((tiles * 32, T, H)); hist_prev = hist.copy(); hist_prev[:, -1] = hist[:, 0]      # frame 0's window when f == 1
            h_prev[:S] = layer0(sd, hist_prev); h_prev = h_prev.reshape(tiles, 32, T, 32, 8).transpose(0, 2, 3, 1, 4)
            for r in (row, row + 16):
                print(f"      cell (tile {tl}, step {t}, k-block {kb}, row {r}): got {hf[tl, t, kb, r]}\n        want {h0[tl, t, kb, r].astype(np.float32)}", flush=True)
                print("        " + whence(hf[tl, t, kb, r], (("this frame's sequence", h0), ("a window of the first row alone", h_prev), ("the bank in front", prev_bank_hf))), flush=True)
        # the ring itself
        ring = fetch(bank, 0, S * n_mc * T * I).reshape(S, n_mc, T, I)
        same_copies = bool((ring == ring[:, :1]).all())
        slots_ok = all(sorted(map(bytes, ring[s, 0])) == sorted(map(bytes, hist[s])) for s in range(S))
        print(f"    ring: copies identical {same_copies}, every stream's slots hold its window's rows {slots_ok}", flush=True)
    m.check()
    n_banks += 1
    del bank, m
print(f"locate: {n_banks} fresh banks, {n_frames} frames, {n_bad} frames off ({time.time() - t0:.0f} s)")
