"""A/B on one box: first-generation cluster kernel with XCD-class cluster formation (default) and with the any-placement form
(diagnostic bit 0x02000000), interleaved, over the shapes that still run on it.  python tests/tools/ab_gen1_classes.py"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
lib = _hip.lib(); stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def model(name):
    cfg = orc.MODEL_CONFIGS[name]
    m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
    m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0))
    return m, cfg
cases = [("uarm", 1024, 64, 0, "cluster_gen1"), ("uarm", 1024, 6, 0, "cluster_gen1"), ("pocket", 512, 64, 0, "cluster_gen1"),
         ("pocket", 300, 6, 0, "cluster_gen1"), ("pocket", 25, 6, _hip.FLAG_DROPOUT_PHILOX, "auto"), ("pocket", 1000, 6, _hip.FLAG_DROPOUT_PHILOX, "cluster"),
         ("watch", 60, 8, _hip.FLAG_DROPOUT_PHILOX, "auto")]
for name, B, T, fl, kern in cases:
    m, cfg = model(name); m.set_kernel(kern)
    x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda"); y2 = torch.empty_like(y)
    def run(flags, out): _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, flags, None, 0.2 if fl else 0.0, 7, C.c_void_p(out.data_ptr()), stream), "fwd")
    for _ in range(40): run(fl, y)
    res = {"classes": [], "any": []}
    for rep in range(5):
        for tag, f, out in (("classes", fl, y), ("any", fl | 0x02000000, y2)):
            for _ in range(10): run(f, out)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
            for _ in range(40): run(f, out)
            b.record(); b.synchronize(); res[tag].append(a.elapsed_time(b) / 40 * 1e3)
    m.check()
    print(f"{name} B={B} T={T} flags={fl:#x} {m.kernel_name(B, T)}: XCD classes {np.median(res['classes']):.1f} us, any placement {np.median(res['any']):.1f} us, "
          f"bit-equal {bool(torch.equal(y, y2))}", flush=True)
    del m
