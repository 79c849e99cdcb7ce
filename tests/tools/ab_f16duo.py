"""configs[4] (watch-only model, fp16 W / x / h): the second-generation kernel (8-member clusters, one workgroup per CU) against its
16-unit-member form (APE_FLAG_ALT_FORM: two workgroups per CU) -- same bits?  per launch?    python tests/tools/ab_f16duo.py [T ...]"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
Ts = [int(a) for a in sys.argv[1:]] or [64, 8]
lib = _hip.lib()
for name in ("watch", "pocket"):
    cfg = orc.MODEL_CONFIGS[name]
    m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0)
    m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
    m.set_norm_stats(np.zeros(cfg["I"]), np.ones(cfg["I"]), np.zeros(cfg["O"]), np.ones(cfg["O"]))
    m.set_precision("f16")
    for B in (1024, 700):
        for T in Ts:
            x = torch.randn(B, T, cfg["I"], device="cuda")
            ys, us, kn = {}, {}, {}
            for form, fl in (("v2", 0), ("duo", _hip.FLAG_ALT_FORM)):
                y = torch.empty(B, cfg["O"], device="cuda")
                def fwd():
                    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT | fl, None, 0.0, 0,
                                                    C.c_void_p(y.data_ptr()), None), "fwd")
                for _ in range(20): fwd()
                torch.cuda.synchronize(); m.check()
                meds = []
                for blk in range(7):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(40): fwd()
                    b.record(); b.synchronize()
                    meds.append(a.elapsed_time(b) / 40 * 1e3)
                m.check()
                ys[form], us[form], kn[form] = y.cpu().numpy().copy(), statistics.median(meds), m.last_kernel()
            d = float(np.abs(ys["v2"] - ys["duo"]).max())
            print(f"{name} {B} x {T:3d}: {kn['v2']} {us['v2']:8.2f} us   {kn['duo']} {us['duo']:8.2f} us   ({(us['duo'] / us['v2'] - 1) * 100:+.1f} %)   "
                  f"max |v2 - duo| {d:.2e}{'  (same bits)' if d == 0 else ''}", flush=True)
