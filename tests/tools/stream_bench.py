"""Per-frame streaming rate of the drop-in estimator classes (reference loop: parse -> window -> model ->
FK -> message).  Reference CPU numbers measured in the survey container: 54 frames/s (mc=60, smooth=5),
586 frames/s (mc=1, smooth=1) -- BASELINE.md section 2."""
import json, shutil, sys, tempfile, time
from array import array
from pathlib import Path
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
import __graft_entry__ as entry; entry.build()
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import config
from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import WatchPhonePocketNN

src = Path(config.PATHS["deploy"]); tmp = Path(tempfile.mkdtemp()) / "deploy"
shutil.copytree(src / "data_stats", tmp / "data_stats")
h = "670b66fa7664252d1cfb3b5a8a362002ffeeba5c"
(tmp / "nn" / h).mkdir(parents=True)
shutil.copy(src / "nn" / h / "results.json", tmp / "nn" / h / "results.json")
sd = orc.make_state_dict(22, 256, 2, 14, 0)
torch.save(({k: torch.from_numpy(v) for k, v in sd.items()}, {}), tmp / "nn" / h / "checkpoint.pt")
config.PATHS["deploy"] = tmp
g = np.load("/root/repo/tests/golden/stream_trace_pocket.npz")
rows = [array("f", r.tolist()) for r in g["rows"]]
for mc, smooth in ((1, 1), (25, 1), (60, 5), (1, 1), (25, 1)):
    est = WatchPhonePocketNN(model_hash=h, smooth=smooth, add_mc_samples=True, monte_carlo_samples=mc)
    def frame(i):
        xx = est.parse_row_to_xx(rows[i % len(rows)])
        pred = est.add_xx_to_row_hist_and_make_prediction(xx)
        return est.msg_from_pred(pred, True)
    for i in range(30): frame(i)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 2000
    tt = np.zeros((n, 4))
    for i in range(n):
        a = time.perf_counter(); xx = est.parse_row_to_xx(rows[i % len(rows)])
        b = time.perf_counter(); pred = est.add_xx_to_row_hist_and_make_prediction(xx)
        c = time.perf_counter(); msg = est.msg_from_pred(pred, True)
        d = time.perf_counter(); tt[i] = (b - a, c - b, d - c, d - a)
    el = time.perf_counter() - t0
    q = lambda c, f: np.quantile(tt[:, c], f) * 1e6
    print(f"mc={mc} smooth={smooth}: {n / el:.0f} frames/s mean; per frame p50 {q(3, .5):.0f} us (parse {q(0, .5):.0f}, window+model "
          f"{q(1, .5):.0f}, fk+msg {q(2, .5):.0f}), p90 {q(3, .9):.0f} us, p99 {q(3, .99):.0f} us; slow stretches: "
          f"{[int(x) for x in (tt[:, 3].reshape(20, -1).mean(1) * 1e6)]}; msg len {len(msg)}")

# ---- stream bank: S streams stepped together, all state on the device (ape_streams_*) ----------------------
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
from wear_mocap_ape_amd.utility import data_stats
from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS
stats = data_stats.get_norm_stats(NNS_INPUTS.WATCH_PHONE_CAL_HIP, NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS)
m = nn_models.DropoutLSTM(22, 256, 2, 14, device=0); m.load_state_dict(sd)
m.set_norm_stats(stats["xx_m"], stats["xx_s"], stats["yy_m"], stats["yy_s"])
raw = torch.from_numpy(g["rows"].astype(np.float32)).cuda()
for S, smooth in ((1, 1), (64, 5), (1024, 1), (1024, 5), (4096, 5)):
    bank = StreamBank(m, S, 6, smooth=smooth, normalize=True, dtype=torch.float32)
    batch = [raw[(torch.arange(S, device="cuda") + f) % len(raw)].contiguous() for f in range(8)]
    for f in range(20):
        bank.push_rows(batch[f % 8], _hip.PARSE_WATCH_PHONE_POCKET); bank.step(with_tail=True)
    torch.cuda.synchronize(); n = 200; t0 = time.perf_counter()
    for f in range(n):
        bank.push_rows(batch[f % 8], _hip.PARSE_WATCH_PHONE_POCKET); bank.step(with_tail=True)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    m.check()
    print(f"stream bank S={S} smooth={smooth}: {el / n * 1e6:.0f} us per frame of all streams = {S * n / el:.0f} stream-frames/s "
          f"({S * n / el / 50:.0f} streams at 50 Hz)")

# ---- the same with Monte-Carlo dropout per stream (ape_streams_set_mc), the reference estimators' default mode -----------
for S, smooth, n_mc in ((1, 1, 25), (64, 5, 25), (1024, 1, 25), (1024, 5, 60), (8192, 1, 25)):
    bank = StreamBank(m, S, 6, smooth=smooth, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc)
    batch = [raw[(torch.arange(S, device="cuda") + f) % len(raw)].contiguous() for f in range(8)]
    for f in range(10):
        bank.push_rows(batch[f % 8], _hip.PARSE_WATCH_PHONE_POCKET); bank.step(with_tail=True)
    torch.cuda.synchronize(); n = 50; t0 = time.perf_counter()
    for f in range(n):
        bank.push_rows(batch[f % 8], _hip.PARSE_WATCH_PHONE_POCKET); bank.step(with_tail=True)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    m.check()
    print(f"stream bank S={S} smooth={smooth} mc={n_mc}: {el / n * 1e6:.0f} us per frame of all streams = {S * n / el:.0f} "
          f"stream-frames/s ({S * n / el / 50:.0f} streams at 50 Hz), {S * n_mc * n / el:.0f} windows/s")
