"""A/B timing of the LSTM kernels (interleaved rounds in one process, HIP events)."""
import ctypes as C, sys, json
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
import __graft_entry__ as entry; entry.build()
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models

name = sys.argv[1] if len(sys.argv) > 1 else "pocket"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = orc.MODEL_CONFIGS[name]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0); m.load_state_dict(sd)
x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
lib = _hip.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
variants = {"cluster": (2, 0), "cluster_noex": (2, 0x40000000), "cluster_noact": (2, 0x20000000),
            "cluster_noex_noact": (2, 0x60000000), "tile16": (1, 0)}
res = {k: [] for k in variants}
def run(kern, flags, n):
    lib.ape_model_set_kernel(m.handle, kern)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, flags, None, 0.0, 0, C.c_void_p(y.data_ptr()), st), "fwd")
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
for k, (kern, fl) in variants.items(): run(kern, fl, 3)
for rnd in range(5):
    for k, (kern, fl) in variants.items():
        res[k].append(run(kern, fl, 10))
# AUTO (B <= 4: small-batch VALU variant)
lib.ape_model_set_kernel(m.handle, 0)
res["auto"] = []
run(0, 0, 3)
for rnd in range(5): res["auto"].append(run(0, 0, 10))
# fp16 variant (configs[4])
lib.ape_model_set_kernel(m.handle, 0); lib.ape_model_set_precision(m.handle, 1)
res["cluster_f16"] = []
run(0, 0, 3)
for rnd in range(5): res["cluster_f16"].append(run(0, 0, 10))
lib.ape_model_set_precision(m.handle, 0)
flop = m.flops_per_window(T) * B
for k, v in res.items():
    med = float(np.median(v))
    print(f"{name} B={B} T={T} {k:20s} median {med:9.1f} us  min {min(v):9.1f} us  {flop / med / 1e6:7.1f} TFLOP/s  per-phase {med / (T + cfg['L'] - 1):6.2f} us")

# in-kernel clock of the cluster kernel (diagnostic stamps): shader cycles / (100 MHz ticks)
import ctypes
raw = ctypes.CDLL(str(_hip.LIB_PATH))
lib.ape_model_set_kernel(m.handle, 2)
for fl, nm in ((0x70000000, "cluster_noex_noact"), (0x10000000, "cluster")):
    for _ in range(200):   # sustained load first
        lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, fl, None, 0.0, 0, C.c_void_p(y.data_ptr()), st)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 14)()
    raw.ape_debug_read_stamps(m.handle, out)
    if out[1]:
        print(f"{nm}: phase loop {out[0]} shader cycles in {out[1] * 10} ns -> in-kernel clock {out[0] / (out[1] * 10):.3f} GHz")
    if any(out[2:14]):
        names = ["fallbacks-l0", "blocking-gather", "mfma", "flags+gather-issue", "act+cell", "barrier-A", "store-issue", "gather-commit", "x-stage", "drain", "barrier-B+flag", "fallbacks-l>0"]
        tot = sum(out[2:14]); P = T + cfg["L"] - 1
        print(f"{nm}: section cycles per phase (wave 0 of workgroup 0): " + ", ".join(f"{n} {v / P:.2f}" if n.startswith("fallb") else f"{n} {v / P:.0f}" for n, v in zip(names, out[2:14])) + f"  | sum {tot / P:.0f}")

# per-workgroup timeline of the last stamped launch (diagnostic library only): cluster = ticket // GH
try:
    buf = (ctypes.c_ulonglong * (256 * 8))()
    if raw.ape_debug_read_wg(m.handle, buf) == 0 and any(buf):
        a = np.array(buf[:], dtype=np.float64).reshape(256, 8)
        GH = cfg["H"] // 16
        t0 = a[:, 2].min()
        cl = (a[:, 0] // GH).astype(int)
        print("cluster: members' XCDs | start spread, prologue end, loop end, kernel end (us after first start)")
        for c in sorted(set(cl)):
            r = a[cl == c]
            print(f"  {c:2d}: xcd {''.join(str(int(v)) for v in r[:, 1])} | start {((r[:, 2].max() - t0) / 100):6.1f} prologue {((r[:, 3].max() - t0) / 100):6.1f}"
                  f" loop {((r[:, 4].min() - t0) / 100):7.1f}..{((r[:, 4].max() - t0) / 100):7.1f} end {((r[:, 5].max() - t0) / 100):7.1f}")
except AttributeError:
    pass
