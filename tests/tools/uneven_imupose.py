"""ImuPoseLSTM's layer-split route (lstm_upper32.hip <32, true> + <32, false>; one-tile clusters on the SOLO form) beside device copies on a
second stream, many launches, every launch against the oracle:   python tests/tools/uneven_imupose.py [B] [T] [launches] [copies per burst] [flags]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _load
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
B, T, N, ncopy = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 1024), (2, 9), (3, 1000), (4, 8)))
flags = int(sys.argv[5], 0) if len(sys.argv) > 5 else 0
DAEMON = _load.start()
sd = orc.make_imupose_state_dict(22, 14, 3)
m = nn_models.ImuPoseLSTM(22, 256, 2, 14, device=0); m.load_state_dict(sd)
x = np.random.default_rng(B * 7 + T).normal(size=(B, T, 22)).astype(np.float32)
xt = torch.from_numpy(x).cuda(); y = torch.empty((B, 14), dtype=torch.float32, device="cuda")
ref = orc.imupose_forward(sd, x)[:, -1]
side = torch.cuda.Stream()
a = torch.full((64 << 20,), 1.0, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
lib = _hip.lib()
off, worst = 0, 0.0
for i in range(N):
    if not DAEMON:
        with torch.cuda.stream(side):
            for _ in range(ncopy): b.copy_(a, non_blocking=True)
    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(xt.data_ptr()), B, T, flags, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
    d = float(np.abs(y.cpu().numpy() - ref).max())
    worst = max(worst, d)
    if d > 2e-6:
        off += 1
        if off <= 5: print(f"launch {i}: {d:.2e}", flush=True)
    side.synchronize()
m.check()
print(f"imupose B={B} T={T} flags={flags:#x} [{m.last_kernel()}]: {off} of {N} launches off, worst {worst:.2e}, aborted checks {m.stats()['aborted_checks']}")
