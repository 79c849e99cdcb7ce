"""VERDICT r05 item 7, measured with a stand-in before anything is built: how long is the eval bank's frame (1024 streams, T = 6) if the NEXT
frame's feature builder runs on a second stream instead of in front of the regressor?  The stand-in keeps the bank's ring out of it (no
race to construct): the main stream steps the bank (regressor + post-filter) on an unchanged ring, a side stream runs `ape_parse_rows` for
1024 rows into a scratch matrix once per frame, issued right behind the step -- the upper bound of what such a pipeline can hide.
python tests/tools/exp_r06_parse_side_stream.py [pocket|watch|uarm]
(mode `ahead`: what was then built on it, ape_streams_push_rows_ahead -- exists at commit 549884f only, measured slower and taken out:
profiles/r06_post_in_tail.md section 3; on any other commit the mode is skipped)"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank

name = sys.argv[1] if len(sys.argv) > 1 else "pocket"
S, frames = 1024, 200
cfg = orc.MODEL_CONFIGS[name]
T = cfg["T"]
kind = {"pocket": _hip.PARSE_WATCH_PHONE_POCKET, "watch": _hip.PARSE_WATCH_ONLY, "uarm": _hip.PARSE_WATCH_PHONE_UARM}[name]
width = _hip.PARSE_SHAPES[kind][0]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
m.set_norm_stats(np.zeros(cfg["I"]), np.ones(cfg["I"]), np.zeros(cfg["O"]), np.ones(cfg["O"]))
m.set_body(orc.DEFAULT_BODY)
rng = np.random.default_rng(3)
rows = [torch.from_numpy(rng.normal(size=(S, width)).astype(np.float32)).cuda() for _ in range(4)]
bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float32)
scratch = torch.empty((S, cfg["I"]), dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
lib = _hip.lib()


def parse_on(stream, f):
    _hip.check(lib.ape_parse_rows(kind, C.c_void_p(rows[f % 4].data_ptr()), S, C.c_void_p(scratch.data_ptr()), _hip.F32,
                                  C.c_void_p(stream.cuda_stream)), "ape_parse_rows")


def run(mode):
    for f in range(T):
        bank.push_rows(rows[f % 4], kind); bank.step_datagrams()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for f in range(frames):
        if mode == "in_order":                    # today's frame: builder -> regressor -> post-filter on one stream
            bank.push_rows(rows[f % 4], kind)
            bank.step_datagrams()
        elif mode == "ahead":                     # the real thing: ape_streams_push_rows_ahead (two window rings, events)
            bank.push_rows(rows[f % 4], kind, ahead=True)
            bank.step_datagrams()
        elif mode == "no_builder":                # the floor: regressor + post-filter only
            bank.step_datagrams()
        else:                                     # the stand-in: the builder of the next frame on the side stream
            bank.step_datagrams()
            parse_on(side, f)
    b.record(); b.synchronize(); torch.cuda.synchronize()
    m.check()
    return a.elapsed_time(b) / frames * 1e3


modes = ("in_order", "no_builder", "side_stream") + (("ahead",) if hasattr(lib, "ape_streams_push_rows_ahead") else ())
for rep in range(3):
    print(f"{name} S={S} T={T}: " + "   ".join(f"{mode} {run(mode):.1f} us" for mode in modes), flush=True)
