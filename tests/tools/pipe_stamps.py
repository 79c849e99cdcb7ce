"""Where a stage of the MLP pipeline kernel spends its time: python tests/tools/pipe_stamps.py [rows]
(wave 0 of pair 0 sums s_memtime -- shader clocks -- over the segments of its tile loop; the last launch).
Runs on the DIAGNOSTIC library (`make -C arm-pose-estimation_amd/csrc diag`): the product library does not read APE_PIPE_DIAG."""
import ctypes as C, os, sys
import numpy as np
os.environ.setdefault("APE_PIPE_DIAG", "8")
os.environ.setdefault("APE_HIP_LIB", "/root/repo/arm-pose-estimation_amd/lib/diag/libape_hip_diag.so")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
m = nn_models.DropoutFF(14, 256, 2, 22, dropout=0.2, device=0)
rng = np.random.default_rng(0)
m.load_weight_blob(torch.from_numpy(rng.uniform(-0.06, 0.06, m.weight_blob_floats()).astype(np.float32)).cuda())
x = torch.randn(N, 1, 22, device="cuda"); y = torch.empty(N, 14, device="cuda"); lib = _hip.lib()
lib.ape_debug_peek_pipe.restype, lib.ape_debug_peek_pipe.argtypes = C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint), C.c_int]
run = lambda: _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), N, 1, 0, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
for _ in range(30): run()
buf = (C.c_uint * 48)()
assert lib.ape_debug_peek_pipe(m.handle, 160, buf, 48) == 0
w = np.frombuffer(buf, dtype=np.uint32).astype(np.uint64)
tiles = (N + 31) // 32 // 128
names = {0: ["barrier", "(top of the iteration)", "layer 1 stream (256 MFMAs)", "leaky_relu + slot look + x staging", "layer 0 stream (32 MFMAs)", "leaky_relu"],
         1: ["copy wait + flag + barrier", "y of tile i - 3", "layer 2 stream (256 MFMAs)", "leaky_relu", "output layer stream (32 MFMAs)"]}
vb = (C.c_uint * 16)()
assert lib.ape_debug_peek_pipe(m.handle, 240, vb, 16) == 0
print("pairs 0..7 share an XCD (producer's verdict):", [int(vb[2 * k]) & 1 for k in range(8)])
for role in (0, 1):
    print(f"stage {'AB'[role]} (pair 0, wave 0; {tiles} tiles):")
    tot = 0
    for k, nm in enumerate(names[role]):
        ticks = int(w[24 * role + 2 * k] | (w[24 * role + 2 * k + 1] << np.uint64(32)))
        tot += ticks
        print(f"  {nm:36s} {ticks / tiles:8.0f} cycles per tile")
    print(f"  {'sum':36s} {tot / tiles:8.0f} cycles per tile   (288 MFMAs alone: 18432)")
