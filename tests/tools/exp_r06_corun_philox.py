"""Round 6, VERDICT r05 item 5: what would generating the NEXT frame's dropout masks on a second stream cost the upper-arm bank's dominant
kernel?  A stand-in: a Philox-bound torch kernel of about the expand kernel's work (39 M bernoulli draws = 19.7 M Philox4x32-10 calls x 2,
written as 39 MB of bytes so that it stays compute-bound) launched on a side stream right before every frame's step.
python tests/tools/exp_r06_corun_philox.py"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
import bench
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.streams import StreamBank
from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS
um = bench._bank_model(bench.UARM, (NNS_INPUTS.WATCH_PHONE_CAL_ALL, NNS_TARGETS.ORI_CAL_LARM_UARM))
width = _hip.PARSE_SHAPES[_hip.PARSE_WATCH_PHONE_UARM][0]
rows = torch.from_numpy(np.random.default_rng(5).normal(size=(1024, width)).astype(np.float32)).cuda()
bank = StreamBank(um, 1024, 6, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=50, dropout=0.2)
side = torch.cuda.Stream()
m = torch.empty(51200 * 6 * 128, dtype=torch.uint8, device="cuda")

def frame(co):
    bank.push_rows(rows, _hip.PARSE_WATCH_PHONE_UARM)
    if co:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.bernoulli_(0.8)
    bank.step_datagrams()

def timed(co, n=30):
    for _ in range(6): frame(co)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): frame(co)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): m.bernoulli_(0.8)
torch.cuda.synchronize(); a.record()
for _ in range(10): m.bernoulli_(0.8)
b.record(); torch.cuda.synchronize()
print(f"stand-in alone: {a.elapsed_time(b) / 10 * 1e3:.1f} us per call (ape_mc_expand128_kernel: ~72 us)")
for rep in range(2):
    print(f"frame alone {timed(False):.1f} us | with the stand-in on a second stream {timed(True):.1f} us")
um.check()
