"""The batch-tile kernel with injected dropout masks against the oracle, many launches, beside the soak tools' continuous device copies
(APE_SOAK_LOAD=1):  python tests/tools/uneven_tile16.py [name] [B] [T] [launches]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import _load
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
name = sys.argv[1] if len(sys.argv) > 1 else "watch"
B, T, reps = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 2460), (3, 8), (4, 60)))
_load.start()
cfg = orc.MODEL_CONFIGS[name]
I, H, L, O = cfg["I"], cfg["H"], cfg["L"], cfg["O"]
sd = orc.make_state_dict(I, H, L, O, 3)
m = nn_models.DropoutLSTM(I, H, L, O, dropout=0.2, device=0); m.load_state_dict(sd); m.set_kernel("tile16")
rng = np.random.default_rng(9)
lib = _hip.lib()
bad = 0
for r in range(reps):
    x = rng.normal(size=(B // 60 + 1, T, I)).astype(np.float32)
    x = np.repeat(x, 60, axis=0)[:B]                       # windows repeated like a bank's sample rows
    masks = [(rng.random((B, T, H)) >= 0.2).astype(np.float32) / np.float32(0.8) for _ in range(L - 1)]
    ref = orc.lstm_forward(sd, x, masks=masks)[:, -1, :]
    xd = torch.from_numpy(x).cuda(); md = torch.from_numpy(np.stack(masks)).cuda()
    y = torch.empty((B, O), dtype=torch.float32, device="cuda")
    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(xd.data_ptr()), B, T, _hip.FLAG_DROPOUT_MASKS, C.c_void_p(md.data_ptr()), 0.2, 0,
                                    C.c_void_p(y.data_ptr()), None), "fwd")
    torch.cuda.synchronize()
    d = np.abs(y.cpu().numpy() - ref).max(axis=1)
    rows = np.nonzero(d > 1e-6)[0]
    if len(rows):
        bad += 1
        print(f"launch {r} [{m.last_kernel()}]: {len(rows)} rows off (max {d.max():.2e}); 60-row groups {sorted(set(int(i) // 60 for i in rows))[:16]}", flush=True)
m.check()
print(f"{name} {B}x{T} batch-tile kernel, injected masks: {bad} of {reps} launches with rows beyond 1e-6")
