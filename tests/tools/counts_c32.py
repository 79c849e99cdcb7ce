"""Light counters of lstm_cluster32.hip in a product-like build (-DAPE_C32_COUNTS: no stamps inside the MFMA stream): per workgroup, how many
gathers took the blocking form, and the shader clocks between a steady-state section's entry and the exit of its top barrier; beside the launch
time by HIP events.  python tests/tools/counts_c32.py [B] [T] [extra flags]      (APE_HIP_LIB = a -DAPE_C32_COUNTS variant)"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
EXTRA = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
cfg = orc.MODEL_CONFIGS["pocket"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0))
m.set_kernel("cluster")
x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
lib = _hip.lib()
lib.ape_debug_read_wg.restype, lib.ape_debug_read_wg.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
def fwd():
    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, EXTRA, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
for _ in range(20): fwd()
meds = []
for blk in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40): fwd()
    b.record(); b.synchronize()
    meds.append(a.elapsed_time(b) / 40 * 1e3)
torch.cuda.synchronize(); m.check()
buf = (C.c_ulonglong * (256 * 8))()
assert lib.ape_debug_read_wg(m.handle, buf) == 0
d = np.frombuffer(buf, dtype=np.uint64)[512:512 + 24 * 8 * 8].reshape(24 * 8, 8).astype(np.float64)
d = d[d[:, 7] > 0]
n = np.maximum(d[:, 6], 1)
print(f"{m.last_kernel()} B={B} T={T} flags {EXTRA:#x}: {statistics.median(meds):.1f} us per launch; last launch, {len(d)} workgroups (wave 0 of each):")
print(f"  blocking gathers per launch   layer 0: mean {d[:, 0].mean():5.2f} max {d[:, 0].max():3.0f}   layer 1: mean {d[:, 1].mean():5.2f} max {d[:, 1].max():3.0f}"
      f"   prefetched: {d[:, 2].mean():5.1f} / {d[:, 3].mean():5.1f}")
print(f"  entry -> behind the top barrier (cycles per steady-state section)   layer 0: mean {(d[:, 4] / n).mean():6.0f} max {(d[:, 4] / n).max():6.0f}"
      f"   layer 1: mean {(d[:, 5] / n).mean():6.0f} max {(d[:, 5] / n).max():6.0f}")
print(f"  kernel body (shader clocks, first section .. head): mean {d[:, 7].mean():9.0f}  min {d[:, 7].min():9.0f} max {d[:, 7].max():9.0f}")
