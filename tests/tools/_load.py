"""A memory-bound neighbour for the soak tools: with APE_SOAK_LOAD=1 in the environment a daemon thread keeps a queue of 256 MiB
device-to-device copies going on a second stream for the life of the process (a few copies in flight at any time), so that everything the
tool launches runs beside a loaded memory side -- the condition that exposed round 4's "cold-start fault" (DESIGN.md 4.17-4.18).
    import _load; _load.start()"""
import os
import threading

_state = {}


def start():
    if os.environ.get("APE_SOAK_LOAD") != "1" or _state:
        return False
    import torch
    side = torch.cuda.Stream()
    a = torch.full((64 << 20,), 1.0, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    stop = threading.Event()

    def run():
        evs = []
        while not stop.is_set():
            with torch.cuda.stream(side):
                b.copy_(a, non_blocking=True)
                e = torch.cuda.Event()
                e.record(side)
            evs.append(e)
            if len(evs) > 6:               # bounded queue: never more than a handful of copies ahead of the device
                evs.pop(0).synchronize()
    th = threading.Thread(target=run, daemon=True)
    th.start()
    _state.update(stop=stop, thread=th, a=a, b=b)
    import atexit

    def _stop():
        stop.set()
        th.join(timeout=5.0)
    atexit.register(_stop)
    print("[load] device copies running on a second stream", flush=True)
    return True
