"""Three estimator threads, each with its own model handle and HIP stream, each launching 64-row batches on the first-generation cluster
kernel in its XCD-class form (8 clusters x 16 members = 128 workgroups per launch: three of them do not fit the chip together).  Every
frame must be the single-threaded result -- bit for bit, or within 1e-6 where ape_model_recover re-issued a launch that gave up -- and no
frame may be lost.  python tests/tools/threads_gen1.py [frames]"""
import sys, threading, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arm-pose-estimation_amd")
import json
import torch
from oracle import ape_oracle as orc
from tests.test_hip_parity import make_model, _synthetic_windows
raw = json.loads(open("/root/repo/tests/golden/norm_stats.json").read())
ns = {k: {kk: np.array(vv) if kk[:2] in ("xx", "yy") else vv for kk, vv in v.items()} for k, v in raw.items()}
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
name, B, NT = "pocket", 64, 3
cfg = orc.MODEL_CONFIGS[name]
models = [make_model(name, 7 + k, ns[name])[0] for k in range(NT)]
for m in models: m.set_kernel("cluster_gen1")
xs = [_synthetic_windows(ns[name], B, 6 + n_frames, cfg["I"], 50 + k) for k in range(NT)]
want = [np.stack([models[k](torch.from_numpy(np.ascontiguousarray(xs[k][:, f:f + 6])), last_step_only=True, normalize_input=True).numpy()[:, 0]
                  for f in range(n_frames)]) for k in range(NT)]
got = [np.empty_like(w) for w in want]
errs = []
def worker(k):
    try:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for f in range(n_frames):
                got[k][f] = models[k](torch.from_numpy(np.ascontiguousarray(xs[k][:, f:f + 6])), last_step_only=True, normalize_input=True).numpy()[:, 0]
    except Exception as exc:
        errs.append((k, repr(exc)))
t0 = time.time()
th = [threading.Thread(target=worker, args=(k,)) for k in range(NT)]
for t in th: t.start()
for t in th: t.join()
print(f"{NT} threads x {n_frames} frames of {B} rows in {time.time() - t0:.1f} s; errors: {errs}")
for k in range(NT):
    st = models[k].stats()
    d = np.abs(got[k] - want[k]).reshape(n_frames, -1).max(axis=1)
    print(f"  thread {k}: stats {st}, frames that differ {int((d > 0).sum())}, worst {float(d.max()):.2e}")
    assert st["lost_calls"] == 0 and float(d.max()) < 1e-6 and (st["reissued_calls"] > 0 or int((d > 0).sum()) == 0)
