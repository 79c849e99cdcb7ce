"""First-generation cluster kernel: XCD-class cluster formation (default from four clusters on) against the any-placement form
(selector bit 0x02000000), by launch size -- queued launches (40 per block) and single host-synchronised launches, interleaved and
rotated.  python tests/tools/exp_r06_gen1_classes_sweep.py [pocket|uarm]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
lib = _hip.lib(); stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
name = sys.argv[1] if len(sys.argv) > 1 else "pocket"
cfg = orc.MODEL_CONFIGS[name]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0))
m.set_kernel("cluster_gen1")
for T in (6, 64):
    for drop in (True, False):
        for B in (48, 64, 96, 128, 192, 256, 384, 512):
            if T == 64 and B not in (64, 128, 256, 512):
                continue
            fl = _hip.FLAG_DROPOUT_PHILOX if drop else 0
            x = torch.randn(B, T, cfg["I"], device="cuda"); y = torch.empty(B, cfg["O"], device="cuda")
            def run(flags): _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, flags, None, 0.2 if drop else 0.0, 7, C.c_void_p(y.data_ptr()), stream), "fwd")
            forms = (("classes", fl), ("any", fl | 0x02000000))
            for _ in range(60):
                for _, f in forms: run(f)
            torch.cuda.synchronize()
            q = {t: [] for t, _ in forms}; s1 = {t: [] for t, _ in forms}
            for rep in range(6):
                for tag, f in (forms if rep % 2 == 0 else forms[::-1]):
                    for _ in range(5): run(f)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
                    for _ in range(30): run(f)
                    b.record(); b.synchronize(); q[tag].append(a.elapsed_time(b) / 30 * 1e3)
            for it in range(40):
                for tag, f in (forms if it % 2 == 0 else forms[::-1]):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); run(f); b.record(); b.synchronize(); s1[tag].append(a.elapsed_time(b) * 1e3)
            m.check()
            print(f"{name} T={T} {'dropout' if drop else 'eval   '} B={B:4d}: queued classes {np.median(q['classes']):6.1f} any {np.median(q['any']):6.1f}   "
                  f"single classes {np.median(s1['classes']):6.1f} any {np.median(s1['any']):6.1f}", flush=True)
