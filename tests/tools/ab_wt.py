"""What does the write-through (sc1) hand-over -- the default since round 5 -- cost each flag-based cluster kernel against the opt-in plain
in-XCD form?  Same process, interleaved blocks of launches with and without APE_FLAG_IN_XCD_PLAIN (include/ape_hip.h):
python tests/tools/ab_wt.py"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank

lib = _hip.lib()

def model(name):
    cfg = orc.MODEL_CONFIGS[name]
    m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
    m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
    m.set_norm_stats(np.zeros(cfg["I"]), np.ones(cfg["I"]), np.zeros(cfg["O"]), np.ones(cfg["O"])); m.set_body(orc.DEFAULT_BODY)
    return m, cfg

def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3

def ab(tag, fn_of_flags, n=20, blocks=5):
    res = {0: [], 1: []}
    for f in (0, 1): timed(lambda: fn_of_flags(f), 5)
    for _ in range(blocks):
        for f in (0, 1): res[f].append(timed(lambda: fn_of_flags(f), n))
    p, w = statistics.median(res[0]), statistics.median(res[1])
    print(f"{tag:46s} plain {p:9.2f} us   write-through {w:9.2f} us   ({(w / p - 1) * 100:+.2f} %)", flush=True)

for name, B, T, prec in (("pocket", 1024, 64, "f32"), ("pocket", 1024, 6, "f32"), ("pocket", 4096, 6, "f32"), ("uarm", 1024, 64, "f32"),
                         ("uarm", 1024, 6, "f32"), ("watch", 1024, 64, "f16"), ("pocket", 600, 6, "f32")):
    m, cfg = model(name)
    if prec == "f16": m.set_precision("f16")
    x = torch.randn(B, T, cfg["I"], device="cuda")
    y = torch.empty(B, cfg["O"], device="cuda")
    def fwd(f):
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT | (0 if f else _hip.FLAG_IN_XCD_PLAIN),
                                        None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
    ab(f"{name} {B} x {T} {prec} [{m.kernel_name(B, T)}]", fwd)
    m.check()
for name, S, n_mc, smooth, kind in (("pocket", 1024, 25, 1, _hip.PARSE_WATCH_PHONE_POCKET), ("watch", 1024, 25, 10, _hip.PARSE_WATCH_ONLY),
                                    ("uarm", 1024, 50, 1, _hip.PARSE_WATCH_PHONE_UARM), ("pocket", 1024, None, 1, _hip.PARSE_WATCH_PHONE_POCKET)):
    m, cfg = model(name)
    T = cfg["T"]
    rows = torch.randn(S, _hip.PARSE_SHAPES[kind][0], device="cuda")
    bank = StreamBank(m, S, T, smooth=smooth, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc, dropout=0.2)
    base = bank._flags
    def frame(f):
        bank._flags = base | (0 if f else _hip.FLAG_IN_XCD_PLAIN)
        bank.push_rows(rows, kind); bank.step_datagrams()
    ab(f"bank {name} S={S} mc={n_mc} T={T} smooth={smooth}", frame, n=8)
    m.check()
    print("   last kernel:", m.last_kernel())
