import ctypes as C, sys, json
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
raw = json.loads(open("/root/repo/tests/golden/norm_stats.json").read())["pocket"]
st = {k: np.array(raw[k]) for k in ("xx_m", "xx_s", "yy_m", "yy_s")}
sd = orc.make_state_dict(22, 256, 2, 14, 8)
m = nn_models.DropoutLSTM(22, 256, 2, 14, device=0); m.load_state_dict(sd)
m.set_norm_stats(st["xx_m"], st["xx_s"], st["yy_m"], st["yy_s"]); m.set_body(orc.DEFAULT_BODY)
lib = _hip.lib()
B, T = 200, 6
lib.ape_model_reserve(m.handle, B)
rng = np.random.default_rng(3)
x = torch.from_numpy((st["xx_m"] + st["xx_s"] * rng.normal(size=(B, T, 22))).astype(np.float32)).cuda()
y_e = torch.empty((B, 14), device="cuda"); y_g = torch.zeros((B, 14), device="cuda")
est_e = torch.empty((B, 21), dtype=torch.float64, device="cuda"); est_g = torch.zeros_like(est_e)
def call(y, est, stream):
    _hip.check(lib.ape_infer(m.handle, C.c_void_p(x.data_ptr()), B, T, 1, C.c_void_p(y.data_ptr()), C.c_void_p(est.data_ptr()), 1, C.c_void_p(stream)), "infer")
call(y_e, est_e, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
y_e2 = y_e.clone(); call(y_e, est_e, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
print("eager repeat equal:", torch.equal(y_e, y_e2))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    call(y_g, est_g, side.cuda_stream); side.synchronize()
    print("side-stream eager equal:", torch.equal(y_g, y_e))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        call(y_g, est_g, side.cuda_stream)
for i in range(4):
    y_g.zero_(); g.replay(); torch.cuda.synchronize()
    d = (y_g - y_e).abs()
    print(f"replay {i}: equal {torch.equal(y_g, y_e)} max diff {d.max().item():.3e} rows differing {(d.max(dim=1).values > 0).sum().item()} first rows {torch.nonzero(d.max(dim=1).values > 0)[:8].flatten().tolist()}")
m.check()
y_ref = orc.infer_windows(sd, st, orc.DEFAULT_BODY, 0, x.cpu().numpy())[0]
print("eager vs oracle", np.abs(y_e.cpu().numpy() - y_ref).max(), "graph vs oracle", np.abs(y_g.cpu().numpy() - y_ref).max())
