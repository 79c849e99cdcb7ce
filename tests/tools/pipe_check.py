"""MLP pipeline kernel against the tile kernel: python tests/tools/pipe_check.py [N ...]"""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from wear_mocap_ape_amd.estimate import nn_models
Ns = [int(a) for a in sys.argv[1:]] or [16384, 16384 + 37, 65536, 262144 + 5]
m = nn_models.DropoutFF(14, 256, 2, 22, dropout=0.2, device=0)
rng = np.random.default_rng(0)
m.load_weight_blob(torch.from_numpy(rng.uniform(-0.06, 0.06, m.weight_blob_floats()).astype(np.float32)).cuda())
for N in Ns:
    x = torch.randn(N, 22, device="cuda")
    m.set_kernel("tile16"); y0 = m(x).cpu().numpy()
    m.set_kernel("auto"); y1 = m(x).cpu().numpy()
    m.check()
    d = np.abs(y0 - y1)
    print(N, "max diff", d.max(), "rows off", int((d.max(axis=1) > 1e-5).sum()), "nan", int(np.isnan(y1).sum()), flush=True)
    if d.max() > 1e-5:
        bad = np.nonzero(d.max(axis=1) > 1e-5)[0]
        print(" first bad rows", bad[:16], " last", bad[-4:], flush=True)
        print(" y0", y0[bad[0]][:4], "y1", y1[bad[0]][:4])
