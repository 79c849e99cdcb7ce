"""Where does the drop-in loop's p99 come from inside bench.py (1.35-1.39 x p50 there, 1.03 x when the loop runs alone in a fresh process)?
The loop before and after the legs that precede it in bench.py's main(): python tests/tools/exp_r06_loop_order.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
import bench
from oracle import ape_oracle as orc
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.utility import data_stats
from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS

P = bench.POCKET
sd = orc.make_state_dict(P["I"], P["H"], P["L"], P["O"], 0)
stats = data_stats.get_norm_stats(NNS_INPUTS.WATCH_PHONE_CAL_HIP, NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS)


def show(tag):
    out = bench.estimator_loop(sd, 1000)
    line = []
    for key in ("mc1_smooth1", "mc25_smooth1", "mc60_smooth5"):
        e = out[key]["device_frame"]
        line.append(f"{key} {e['p50_us']:.1f}/{e['p99_us']:.1f} ({e['p99_us'] / e['p50_us']:.2f}; launch calls p99 {e['split_us']['launch_calls']['p99']:.1f})")
    print(f"{tag}: " + "  ".join(line), flush=True)


show("fresh process")
model = nn_models.DropoutLSTM(P["I"], P["H"], P["L"], P["O"], device=0)
model.load_state_dict(sd)
model.set_norm_stats(stats["xx_m"], stats["xx_s"], stats["yy_m"], stats["yy_s"])
model.set_body(orc.DEFAULT_BODY)
x = torch.randn(1024, 64, P["I"], device="cuda")
for _ in range(300):
    y = model(x, last_step_only=True)
torch.cuda.synchronize()
show("behind 300 headline steps")
b1 = bench.batch1_latency(model, stats)
print("batch1", round(b1["p50_us"], 1), round(b1["p99_us"], 1), flush=True)
show("behind batch1_latency")
show("again")
