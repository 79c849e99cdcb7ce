"""Do the plain-store (in-L2) hand-overs of the older flag-based kernels survive cold address translations?  Round 4's stale-slice fault on
lstm_upper128.hip showed only on the first launch of a fresh process; one reading is that a plain store's acknowledgement can come back
while its address translation is still being walked, so that the flag (another page, already translated) overtakes the slice.  This tool
evicts the translation caches between launches -- a pass over a buffer of several GB with one touch per 4 KB page, on every CU -- and holds
each launch to the batch-tile kernel's result:   python tests/tools/tlb_stress.py [rounds] [GiB]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
lib = _hip.lib()
big = torch.zeros(int(gib * (1 << 30)) // 4, dtype=torch.float32, device="cuda")
pages = big.view(-1, 1024)                                   # one row = one 4 KB page
def thrash():
    pages[:, 0].add_(1.0)                                    # one touch per page, strided: every access a fresh translation
cases = []
for name, B, T, prec in (("pocket", 1024, 6, "f32"), ("pocket", 1024, 64, "f32"), ("uarm", 1024, 6, "f32"), ("uarm", 1024, 64, "f32"), ("watch", 1024, 64, "f16")):
    cfg = orc.MODEL_CONFIGS[name]
    m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0)
    m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
    m.set_norm_stats(np.zeros(cfg["I"]), np.ones(cfg["I"]), np.zeros(cfg["O"]), np.ones(cfg["O"]))
    x = torch.randn(B, T, cfg["I"], device="cuda")
    def run(flags=0, m=m, x=x, B=B, T=T, cfg=cfg):
        y = torch.empty(B, cfg["O"], device="cuda")
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT | flags, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
        torch.cuda.synchronize()
        return y.cpu().numpy()
    m.set_kernel("tile16"); ref = run(); m.set_kernel("auto")
    if prec == "f16": m.set_precision("f16")
    warm = run()
    cases.append((name, B, T, prec, m, run, ref, warm))
for name, B, T, prec, m, run, ref, warm in cases:
    tol = 5e-3 if prec == "f16" else 2e-5
    off = 0; worst = 0.0; notbits = 0
    for r in range(rounds):
        thrash(); torch.cuda.synchronize()
        y = run()
        d = float(np.abs(y - ref).max()); worst = max(worst, d)
        if d > tol: off += 1
        if not np.array_equal(y, warm): notbits += 1
    m.check()
    print(f"{name} {B} x {T} {prec} {m.last_kernel():28s}: {off} of {rounds} launches behind a translation-cache flush off (worst {worst:.2e}); "
          f"{notbits} not bit-equal to the warm launch", flush=True)
