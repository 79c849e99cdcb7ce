"""one launch of the Monte-Carlo latency kernel, then the health check (which names the wait that gave up): python tests/tools/one_mc_launch.py [n]"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
cfg = orc.MODEL_CONFIGS["pocket"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0); m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0))
x = torch.randn(1, 6, cfg["I"], device="cuda"); y = torch.empty(n, cfg["O"], device="cuda")
lib = _hip.lib()
for it in range(3):
    t0 = time.perf_counter()
    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), n, 6, _hip.FLAG_DROPOUT_PHILOX | _hip.FLAG_BROADCAST_X, None, 0.2, it, C.c_void_p(y.data_ptr()), None), "fwd")
    torch.cuda.synchronize()
    print(f"launch {it}: {(time.perf_counter() - t0) * 1e3:.2f} ms, kernel {m.last_kernel()}", flush=True)
    try:
        m.check()
    except UserWarning as e:
        print("check:", e, flush=True)
