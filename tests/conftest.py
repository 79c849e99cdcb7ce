import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
GOLDEN = REPO / "tests" / "golden"
# the product package lives in a src-style directory, like the reference's src/wear_mocap_ape
for p in (str(REPO), str(REPO / "arm-pose-estimation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(GOLDEN / name, allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def norm_stats():
    import json
    raw = json.loads((GOLDEN / "norm_stats.json").read_text())
    return {k: {kk: np.array(vv) if kk[:2] in ("xx", "yy") else vv for kk, vv in v.items()} for k, v in raw.items()}


@pytest.fixture(scope="session", autouse=True)
def _memory_bound_neighbour():
    """APE_SOAK_LOAD=1 python -m pytest tests -m gpu ...: the whole GPU suite beside a queue of 256 MiB device copies on a second stream for
    the life of the session (tests/tools/_load.py) -- every parity assertion then holds under a loaded memory side, the condition that
    exposed the two inline-asm faults of round 5.  Timing assertions are not meant for this mode.  Off by default."""
    if os.environ.get("APE_SOAK_LOAD") == "1" and _has_gpu():
        sys.path.insert(0, str(REPO / "tests" / "tools"))
        import _load
        _load.start()
    yield
