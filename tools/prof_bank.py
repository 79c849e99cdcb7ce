"""One Monte-Carlo stream-bank configuration, a few frames, for rocprofv3 --kernel-trace --stats:
python3 tools/prof_bank.py [S] [n_mc] [frames]"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import torch
import __graft_entry__ as entry; entry.build()
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.streams import StreamBank, flatten_state_dict
from wear_mocap_ape_amd.utility import data_stats
from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 25
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
stats = data_stats.get_norm_stats(NNS_INPUTS.WATCH_PHONE_CAL_HIP, NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS)
m = nn_models.DropoutLSTM(22, 256, 2, 14, device=0)
rng = np.random.default_rng(0)
n = m.weight_blob_floats()
m.load_weight_blob(torch.from_numpy(rng.uniform(-1 / 16, 1 / 16, n).astype(np.float32)).cuda())
m.set_norm_stats(stats["xx_m"], stats["xx_s"], stats["yy_m"], stats["yy_s"])
g = np.load("/root/repo/tests/golden/stream_trace_pocket.npz")
raw = torch.from_numpy(g["rows"].astype(np.float32)).cuda()
bank = StreamBank(m, S, 6, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc)
batch = [raw[(torch.arange(S, device="cuda") + f) % len(raw)].contiguous() for f in range(8)]
for f in range(5):
    bank.push_rows(batch[f % 8], _hip.PARSE_WATCH_PHONE_POCKET); bank.step_datagrams()
torch.cuda.synchronize(); t0 = time.perf_counter()
for f in range(frames):
    bank.push_rows(batch[f % 8], _hip.PARSE_WATCH_PHONE_POCKET); bank.step_datagrams()
torch.cuda.synchronize(); el = time.perf_counter() - t0
m.check()
print(f"S={S} n_mc={n_mc}: {el / frames * 1e3:.3f} ms per frame of all streams, {S * n_mc * frames / el / 1e6:.2f} M sample windows/s")
