"""Round 6: the one-agent instantiation of ape_lstm_level16 for the 2 x 256 models (pocket, watch-only) against the first generation and the oracle,
5 .. 256 rows: python tests/tools/exp_r06_lv16_pocket.py"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
lib = _hip.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name in ("pocket", "watch"):
    cfg = orc.MODEL_CONFIGS[name]
    sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
    m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0); m.load_state_dict(sd)
    for B in (5, 16, 64, 128, 256, 257):
        for T in (6, 12, 64):
            x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
            ref = orc.lstm_forward(sd, x.cpu().numpy())[:, -1]
            out = []
            for kern in ("auto", "cluster_gen1"):
                m.set_kernel(kern)
                run = lambda: _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, 0, None, 0.0, 0, C.c_void_p(y.data_ptr()), st), "fwd")
                for _ in range(20): run()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
                for _ in range(50): run()
                b.record(); b.synchronize()
                m.check()
                out.append((m.last_kernel(), a.elapsed_time(b) / 50 * 1e3, float(np.abs(y.cpu().numpy() - ref).max())))
            print(f"{name} B={B} T={T}: " + "  |  ".join(f"{k}: {us:7.1f} us (max|dy| {e:.1e})" for k, us, e in out), flush=True)
    m.set_kernel("auto")
