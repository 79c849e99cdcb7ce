"""Large batches of short windows (the stream bank's shape: S*n_mc rows, T=6), cluster vs tile16, with and without dropout:
python tests/tools/time_big_batch.py [B] [T]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = orc.MODEL_CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "pocket"]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0); m.load_state_dict(sd)
x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
lib = _hip.lib()
for kern in ("auto", "cluster", "tile16"):
    lib.ape_model_set_kernel(m.handle, {"auto": 0, "tile16": 1, "cluster": 2}[kern])
    for flags, p, tag in ((0, 0.0, "eval"), (_hip.FLAG_DROPOUT_PHILOX, 0.2, "philox dropout")):
        run = lambda: _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, flags, None, p, 123, C.c_void_p(y.data_ptr()), None), "fwd")
        for _ in range(5): run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): run()
        b.record(); b.synchronize(); m.check()
        us = a.elapsed_time(b) * 100
        print(f"{kern:8s} {tag:15s} B={B} T={T}: {us:9.1f} us  {B / us:7.2f} M windows/s  {m.flops_per_window(T) * B / us / 1e6:6.1f} TFLOP/s")
