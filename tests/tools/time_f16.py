"""fp16 cluster kernels side by side (HIP events, medians): first generation, second generation in its in-L2 form
(when the clusters turn out XCD-pure) and with the any-placement write-through exchange forced.
python tests/tools/time_f16.py [watch|pocket] [B] [T]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models

name = sys.argv[1] if len(sys.argv) > 1 else "watch"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = orc.MODEL_CONFIGS[name]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0); m.load_state_dict(sd)
x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
lib = _hip.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(n, flags):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, flags, None, 0.0, 0, C.c_void_p(y.data_ptr()), st), "fwd")
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3

flop = m.flops_per_window(T) * B
ref = orc.lstm_forward(sd, x.cpu().numpy()[:8], storage="f16")[:, -1]
for prec, flags, tag in (("f32", 0, "f32 cluster kernel"), ("f16_gen1", 0, "fp16 gen 1"), ("f16", 0, "fp16 gen 2 (in-L2 if XCD-pure)"),
                         ("f16", 0x08000000, "fp16 gen 2, write-through forced")):
    m.set_precision(prec)
    run(40, flags)
    v = [run(20, flags) for _ in range(9)]
    err = float(np.abs(y.cpu().numpy()[:8] - ref).max())
    m.check()
    print(f"{tag:36s} {m.kernel_name(B, T):28s} {name} B={B} T={T}: median {np.median(v):8.1f} us  min {min(v):8.1f}  "
          f"{flop / np.median(v) / 1e6:7.1f} TFLOP/s  max|dy vs f16 oracle| {err:.1e}", flush=True)
