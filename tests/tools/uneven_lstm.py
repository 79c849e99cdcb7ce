"""One LSTM kernel beside a queue of device copies on a second stream, every launch against the oracle, with the rows that are off located:
python tests/tools/uneven_lstm.py <pocket|watch|uarm> <B> <T> <f32|f16|f16_gen1> <auto|cluster|cluster_gen1> [launches] [copies per launch] [extra flags] [state-dict tensors to zero, comma separated]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
name, B, T, prec, kernel = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
ncopy = int(sys.argv[7]) if len(sys.argv) > 7 else 24
extra = int(sys.argv[8], 0) if len(sys.argv) > 8 else 0
ZERO = sys.argv[9].split(",") if len(sys.argv) > 9 else []
cfg = orc.MODEL_CONFIGS[name]
raw = json.loads(open(os.path.join(ROOT, "tests", "golden", "norm_stats.json")).read())[name]
st = {k: np.array(raw[k]) for k in ("xx_m", "xx_s", "yy_m", "yy_s")}
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 13)
for key in ZERO: sd[key][:] = 0.0
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0)
m.load_state_dict(sd); m.set_norm_stats(st["xx_m"], st["xx_s"], st["yy_m"], st["yy_s"])
if prec != "f32": m.set_precision(prec)
m.set_kernel(kernel)
x = (st["xx_m"] + st["xx_s"] * np.random.default_rng(B + T).normal(size=(B, T, cfg["I"]))).astype(np.float32)
if os.environ.get("CONSTX"): x[:] = x[:, :1]              # the same row at every step: a wrong STEP of x is then invisible, a stale h slice is not
xn = ((x.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
ref = orc.lstm_forward(sd, xn, storage="f16")[:, -1] if prec != "f32" else orc.lstm_forward(sd, xn)[:, -1]
tol = 3e-4 if prec != "f32" else 1e-6
xd = torch.from_numpy(x).cuda()
lib = _hip.lib()
side = torch.cuda.Stream()
a = torch.full((64 << 20,), 1.0, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
bad_launches = 0
for r in range(reps):
    y = torch.empty(B, cfg["O"], device="cuda")
    torch.cuda.synchronize()
    if ncopy:
        with torch.cuda.stream(side):
            for _ in range(ncopy): b.copy_(a, non_blocking=True)
    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(xd.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT | extra, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
    torch.cuda.synchronize()
    d = np.abs(y.cpu().numpy() - ref).max(axis=1)
    bad = np.nonzero(d > tol)[0]
    if len(bad):
        bad_launches += 1
        cl = sorted(set(int(i) // 32 for i in bad))
        print(f"launch {r} [{m.last_kernel()}]: {len(bad)} rows off (max {d.max():.2e}); 32-row groups: {cl[:24]}", flush=True)
m.check()
print(f"{name} {B}x{T} {prec} {kernel} [{m.last_kernel()}] flags {extra:#x} zeroed {ZERO}: {bad_launches} of {reps} launches with rows beyond {tol:g} (copies per launch: {ncopy}); aborted: {m.stats()['aborted_checks']}")
