"""Asserting run of the 3 x 128 Monte-Carlo bank's upper-layer kernel (lstm_upper128.hip built with -DAPE_UP128_ASSERT): which bytes does a
section find in its gathered operands?  Weights under which a layer's fresh h is ONE number per step (W = 0, per-gate constant biases), so every
float of a gathered slice must equal what the checking lane itself computed in the set's section in front; a mismatch is classified by where
its value sits in the layer's sequence: an OLDER step (the slice had not arrived: visibility / a copy counted as landed too early), a NEWER
step (overwritten by a fast producer: protocol), zero (never written).  The process's FIRST launch is the one that counts.
    APE_HIP_LIB=.../lib/ab/libape_<variant>.so python tests/tools/assert_up128.py [S] [n_mc] [frames] [device copies per frame on a second stream]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-pose-estimation_amd"))
import numpy as np
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 50
FRAMES = int(sys.argv[3]) if len(sys.argv) > 3 else 3
NCOPY = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cfg = orc.MODEL_CONFIGS["uarm"]
T, H = cfg["T"], cfg["H"]
sd = orc.make_state_dict(cfg["I"], H, cfg["L"], cfg["O"], 5)
for l, gate_bias in ((1, (0.3, 0.5, 0.9, 0.7)), (2, (0.6, 0.4, 0.8, 0.2))):        # layers 1 and 2: no weights, one bias per gate (i, f, g, o)
    sd[f"lstm.weight_ih_l{l}"][:] = 0.0
    sd[f"lstm.weight_hh_l{l}"][:] = 0.0
    sd[f"lstm.bias_hh_l{l}"][:] = 0.0
    sd[f"lstm.bias_ih_l{l}"][:] = np.repeat(np.array(gate_bias, dtype=np.float32), H)
m = nn_models.DropoutLSTM(cfg["I"], H, cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(sd)
m.set_norm_stats(np.zeros(cfg["I"]), np.ones(cfg["I"]), np.zeros(cfg["O"]), np.ones(cfg["O"])); m.set_body(orc.DEFAULT_BODY)
lib = _hip.lib()
lib.ape_debug_read_wg.restype, lib.ape_debug_read_wg.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
rows = torch.randn(S, 55, device="cuda")
bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc, dropout=0.2)
f32 = lambda bits: float(np.array([bits], dtype=np.uint32).view(np.float32)[0])
side = torch.cuda.Stream()
if NCOPY:
    ca = torch.full((64 << 20,), 1.0, dtype=torch.float32, device="cuda"); cb = torch.empty_like(ca)
last_n = 0
for frame in range(FRAMES):                              # frame 0 = the process's cold launch
    bank.push_rows(rows, _hip.PARSE_WATCH_PHONE_UARM)
    torch.cuda.synchronize()
    if NCOPY:
        with torch.cuda.stream(side):
            for _ in range(NCOPY): cb.copy_(ca, non_blocking=True)
    bank.step_datagrams()
    torch.cuda.synchronize(); m.check()
    assert m.last_kernel() == "ape_lstm_upper128", m.last_kernel()
    buf = (C.c_ulonglong * (256 * 8))()
    assert lib.ape_debug_read_wg(m.handle, buf) == 0
    d = np.frombuffer(buf, dtype=np.uint64)
    n = int(d[1024]) & 0xFFFFFFFF
    seq = {0: [], 2: []}                                 # the layers' sequences by section of a set: h_1 of step k, h_2 of step k - 1
    for k in range(2 * T + 1):
        seq[0].append(int(d[1600 + 2 * k]) & 0xFFFFFFFF); seq[2].append(int(d[1601 + 2 * k]) & 0xFFFFFFFF)
    tag = "COLD " if frame == 0 else "warm "
    print(f"{tag}frame {frame}: {n - last_n} new mismatching (lane, section, kind) records (cumulative {n})" + ("" if n else "  -- every gathered float was the expected one"))
    last_n = n
    by = {}
    for r in range(min(n, 120)):
        o = d[1032 + 4 * r: 1036 + 4 * r]
        cl, mem, s, k = int(o[0] >> 48), int(o[0] >> 40) & 0xFF, int(o[0] >> 32) & 0xFF, int(o[0]) & 0xFFFFFFFF
        phase, kind, kb, lane = int(o[1] >> 48), int(o[1] >> 32) & 0xFFFF, int(o[1] >> 8) & 0xFFFFFF, int(o[1]) & 0xFF
        found, expect = int(o[2] >> 32), int(o[2]) & 0xFFFFFFFF
        table = [f32(b) for b in (seq[2] if kind == 2 else seq[0])]
        fv, ev = (f32(found) / 1.25 if kind == 1 else f32(found)), f32(expect)
        near = lambda x: next((i for i, tv in enumerate(table) if tv != 0.0 and abs(tv - x) <= 2e-6 * abs(tv)), None)
        if found == 0: what = "zero (never written)"
        elif near(fv) is not None and near(ev) is not None:
            dlt = near(fv) - near(ev)
            what = f"{f32(found):.6f} = the value of {abs(dlt)} section(s) {'EARLIER: not arrived yet' if dlt < 0 else 'LATER: overwritten'} (expected {ev:.6f})"
        else: what = f"{f32(found):.6f} (not a value of this layer's sequence; expected {ev:.6f})"
        key = (cl, mem, s, k, kind, phase)
        by.setdefault(key, []).append((kb, lane, what, int(o[3])))
    for key in sorted(by)[:40]:
        cl, mem, s, k, kind, phase = key
        v = by[key]
        print(f"   cluster {cl} member {mem} set {s} section {k} kind {('h_1', 'h_1 masked', 'h_2')[kind]} phase {phase} ({'behind the top barrier' if phase == 0 else 'end of the section'}): "
              f"{len(v)} lanes, first k-block {min(x[0] for x in v)}, e.g. {v[0][2]}")
    # the counter is cumulative over launches: zero it by reading differences
    if frame == 0: first_n = n
