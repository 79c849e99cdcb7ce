"""Sustained run of the Monte-Carlo stream bank on the weight-stationary route: many frames back to back, a health check every
`every` frames (an aborted launch would show as aborted_checks / reissued_calls in ape_model_stats), a checksum of the messages
to show the outputs stay finite.  python tests/tools/stability_bank.py [S] [n_mc] [frames] [every]"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
from oracle import ape_oracle as orc
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 25
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
every = int(sys.argv[4]) if len(sys.argv) > 4 else 500
cfg = orc.MODEL_CONFIGS["pocket"]
m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
m.load_state_dict(orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5))
m.set_norm_stats(np.zeros(22), np.ones(22), np.zeros(14), np.ones(14)); m.set_body(orc.DEFAULT_BODY)
rows = [torch.randn(S, 55, device="cuda") for _ in range(8)]
bank = StreamBank(m, S, 6, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc, dropout=0.2)
t0 = time.time(); bad = 0
for f in range(frames):
    bank.push_rows(rows[f % 8], _hip.PARSE_WATCH_PHONE_POCKET)
    out = bank.step_datagrams()
    if (f + 1) % every == 0:
        bank.recover()
        fin = bool(torch.isfinite(out).all().item())
        bad += 0 if fin else 1
        print(f"  frame {f + 1}: {time.time() - t0:.0f} s, stats {m.stats()}, finite {fin}", flush=True)
st = m.stats()
print(f"S={S} n_mc={n_mc}: {frames} frames in {time.time() - t0:.0f} s, stats {st}, non-finite checks {bad}")
assert st["aborted_checks"] == 0 and st["lost_calls"] == 0 and bad == 0
