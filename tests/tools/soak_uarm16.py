"""Soak of the upper-arm model's second-generation kernel (lstm_cluster16.hip): random batch sizes and window lengths in its dispatch range,
fused normalisation on / off, both exchange forms, against the batch-tile kernel; launches back to back without a sync in between (the
self-cleaning control words must be ready for the next launch).  python tests/tools/soak_uarm16.py [seconds]"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
cfg = orc.MODEL_CONFIGS["uarm"]
rng = np.random.default_rng(5)
lib = _hip.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
t0 = time.time(); n = 0; worst = 0.0
while time.time() - t0 < budget:
    m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
    m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], int(rng.integers(1000))))
    m.set_norm_stats(rng.normal(size=cfg["I"]), rng.uniform(0.5, 2.0, size=cfg["I"]), np.zeros(cfg["O"]), np.ones(cfg["O"]))
    for _ in range(12):
        B, T = int(rng.integers(513, 2600)), int(rng.choice([12, 13, 14, 15, 16, 31, 64, 77]))
        flags = (_hip.FLAG_NORMALIZE_INPUT if rng.integers(2) else 0) | (0x08000000 if rng.integers(3) == 0 else 0)
        x = torch.randn(B, T, cfg["I"], device="cuda")
        ys = [torch.empty(B, cfg["O"], device="cuda") for _ in range(3)]
        m.set_kernel("cluster")
        assert m.kernel_name(B, T) == "ape_lstm_cluster16<128, 3, 64, 2>"
        for y in ys:                                     # three launches back to back
            _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, flags, None, 0.0, 0, C.c_void_p(y.data_ptr()), st), "fwd")
        m.check()
        y0 = torch.empty(B, cfg["O"], device="cuda")
        m.set_kernel("tile16")
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, flags & ~0x08000000, None, 0.0, 0, C.c_void_p(y0.data_ptr()), st), "fwd")
        m.check()
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]), (B, T, flags)
        d = float((ys[0] - y0).abs().max())
        worst = max(worst, d)
        assert d < 1e-6, (B, T, flags, d)
        n += 1
    del m
    print(f"  ... {n} calls, {time.time() - t0:.0f} s, worst |cluster16 - batch-tile| {worst:.2e}", flush=True)
print(f"uarm cluster16 soak: {n} random calls x 3 launches in {time.time() - t0:.0f} s, worst {worst:.2e}, all repeat launches bit-equal")
