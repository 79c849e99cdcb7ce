"""A Monte-Carlo bank under injected masks against the oracle, frame by frame, beside a queue of device copies on a second stream, with the
rows that are off located (tile, cluster, set):
APE_HIP_LIB=.../lib/diag/libape_hip_testhooks.so python tests/tools/uneven_bank.py [S] [n_mc] [frames] [copies per burst] [pocket|watch|uarm] [smooth]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 50
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ncopy = int(sys.argv[4]) if len(sys.argv) > 4 else 24
smooth = int(sys.argv[6]) if len(sys.argv) > 6 else 1
name = sys.argv[5] if len(sys.argv) > 5 else "uarm"
cfg = orc.MODEL_CONFIGS[name]
raw = json.loads(open(os.path.join(ROOT, "tests", "golden", "norm_stats.json")).read())[name]
st = {k: np.array(raw[k]) for k in ("xx_m", "xx_s", "yy_m", "yy_s")}
T, I, O, H, L = cfg["T"], cfg["I"], cfg["O"], cfg["H"], cfg["L"]
sd = orc.make_state_dict(I, H, L, O, 5)
m = nn_models.DropoutLSTM(I, H, L, O, dropout=0.2, device=0)
m.load_state_dict(sd); m.set_norm_stats(st["xx_m"], st["xx_s"], st["yy_m"], st["yy_s"]); m.set_body(orc.DEFAULT_BODY)
lib = _hip.lib()
lib.ape_debug_set_bank_masks.restype, lib.ape_debug_set_bank_masks.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
lib.ape_debug_bank_targets.restype, lib.ape_debug_bank_targets.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _load
if _load.start(): ncopy = 0          # APE_SOAK_LOAD=1: a thread keeps copies in flight the whole time instead of the bursts
side = torch.cuda.Stream()
a = torch.full((64 << 20,), 1.0, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
rows = S * n_mc
rng = np.random.default_rng(S * 1000 + n_mc)
feats = (st["xx_m"] + st["xx_s"] * np.random.default_rng(8).normal(size=(S, frames, I))).astype(np.float32)
NORM = os.environ.get('NORM', '1') == '1'
bank = StreamBank(m, S, T, smooth=smooth, normalize=NORM, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=3)
hist = [[] for _ in range(S)]
n_tiles = (rows + 31) // 32
NC = min(64, (n_tiles + 7) // 8 * 8)
for f in range(frames):
    masks = [(rng.random((rows, T, H)) >= 0.2).astype(np.float32) / np.float32(0.8) for _ in range(L - 1)]
    md = torch.from_numpy(np.stack(masks)).cuda()
    assert lib.ape_debug_set_bank_masks(bank._handle, C.c_void_p(md.data_ptr())) == 0
    bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda())
    torch.cuda.synchronize()
    if ncopy:
        with torch.cuda.stream(side):
            for _ in range(ncopy): b.copy_(a, non_blocking=True)
    bank.step()
    y = np.empty((rows, O), dtype=np.float32)
    assert lib.ape_debug_bank_targets(bank._handle, y.ctypes.data_as(C.c_void_p)) == 0
    side.synchronize()
    wins = []
    for s in range(S):
        hist[s].append(feats[s, f])
        while len(hist[s]) < T: hist[s].append(feats[s, f])
        del hist[s][:len(hist[s]) - T]
        xn = ((np.stack(hist[s]).astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32) if NORM else np.stack(hist[s]).astype(np.float32)
        wins.append(np.repeat(xn[None], n_mc, axis=0))
    ref = orc.lstm_forward(sd, np.concatenate(wins), masks=masks)[:, -1, :]
    d = np.abs(y - ref).max(axis=1)
    bad = np.nonzero(d > 1e-6)[0]
    if len(bad):
        tiles = sorted(set(int(r) // 32 for r in bad))
        info = [(t, t % NC, (t // NC) % 2, t // (2 * NC), int((d[t * 32:(t + 1) * 32] > 1e-6).sum())) for t in tiles]
        print(f"frame {f} [{m.last_kernel()}]: {len(bad)} rows off (max {d.max():.2e}); (tile, cluster, set, index in set, rows off): {info[:16]}", flush=True)
    else:
        print(f"frame {f} [{m.last_kernel()}]: ok (max {d.max():.2e})", flush=True)
m.check()
print("stats", m.stats())
