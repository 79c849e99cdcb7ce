"""Launch time of the MLP regressor kernel (DropoutFF), HIP events: python tests/tools/time_mlp.py [B] [hidden_layers] [auto|tile16]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_hidden = int(sys.argv[2]) if len(sys.argv) > 2 else 2
m = nn_models.DropoutFF(14, 256, n_hidden, 22, dropout=0.2, device=0)      # (output, hidden, count, input): the reference's order
if len(sys.argv) > 3: m.set_kernel(sys.argv[3])
rng = np.random.default_rng(0)
m.load_weight_blob(torch.from_numpy(rng.uniform(-0.06, 0.06, m.weight_blob_floats()).astype(np.float32)).cuda())
x = torch.randn(B, 1, 22, device="cuda"); y = torch.empty(B, 14, device="cuda"); lib = _hip.lib()
run = lambda: _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, 1, 0, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
for _ in range(20): run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
for _ in range(200): run()
b.record(); b.synchronize()
us = a.elapsed_time(b) / 200 * 1e3
print(f"DropoutFF(22,256,{n_hidden},14) B={B}: {us:.1f} us per launch, {B / us:.2f} M rows/s, {m.flops_per_window(1) * B / us / 1e6:.2f} TFLOP/s")
