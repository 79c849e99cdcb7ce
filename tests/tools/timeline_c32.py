"""Timeline of one short launch of lstm_cluster32.hip (diagnostic build): cluster 0 / member 0 / thread 0, microseconds from kernel entry.
APE_HIP_LIB=.../libape_hip_diag.so python tests/tools/timeline_c32.py [T] [MHz]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mhz = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0        # s_memtime / readcyclecounter tick rate: found from the launch's event time
cfg = orc.MODEL_CONFIGS["pocket"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0))
m.set_kernel("cluster")
x = torch.randn(1024, T, cfg["I"], device="cuda")
for _ in range(20): m(x, last_step_only=True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): m(x, last_step_only=True)
b.record(); b.synchronize(); m.check()
us = a.elapsed_time(b) / 20 * 1e3
lib = _hip.lib(); buf = (C.c_ulonglong * 2048)()
lib.ape_debug_read_wg.argtypes = [C.c_void_p, C.c_void_p]
lib.ape_debug_read_wg(m.handle, buf)
d = np.frombuffer(buf, dtype=np.uint64)[256:456].astype(np.float64)
t0 = d[0]
span = d[5] - t0
print(f"T={T}: launch {us:.1f} us by events (diagnostic build); entry -> head done {span:.0f} ticks")
tick = 1.0 / mhz
def at(i): return (d[i] - t0) * tick
print(f"  rendezvous done {at(1):7.2f}   step 0 + weights {at(2):7.2f}")
for ph in range(1, T + 1):
    for l in range(2):
        i = 8 + 2 * (2 * ph + l)
        if d[i] > 0: print(f"  phase {ph} layer {l} (step {ph - l:2d}): behind top barrier {at(i):7.2f}   end {at(i + 1):7.2f}   (top wait {at(i) - prev if 'prev' in dir() else 0:5.2f}, body {at(i + 1) - at(i):5.2f})")
        prev = at(i + 1) if d[i] > 0 else prev if 'prev' in dir() else 0
print(f"  final gather done {at(4):7.2f}   head done {at(5):7.2f}")
