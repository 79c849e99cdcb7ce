"""Is the tail of the drop-in Estimator loop (bench.py batch1.estimator_loop, p99 / p50 at Monte-Carlo settings) the interpreter's garbage
collector?  The same loop with the collector on (as measured so far) and off (what `timeit` does while it times): python tests/tools/exp_r06_loop_gc.py"""
import gc
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from oracle import ape_oracle as orc

cfg = orc.MODEL_CONFIGS["pocket"]
sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
for rep in range(2):
    for mode in ("gc on", "gc off"):
        if mode == "gc off":
            gc.collect(); gc.disable()
        else:
            gc.enable()
        out = bench.estimator_loop(sd, 2000)
        gc.enable()
        line = []
        for key in ("mc1_smooth1", "mc25_smooth1", "mc60_smooth5"):
            for form in ("device_frame", "device_frame_array"):
                e = out.get(key, {}).get(form)
                if e:
                    line.append(f"{key}/{form[13:] or 'list'} {e['p50_us']:.1f}/{e['p99_us']:.1f} ({e['p99_us'] / e['p50_us']:.2f})")
        print(mode + ": " + "  ".join(line), flush=True)
