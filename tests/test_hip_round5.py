"""Round-5 GPU tests: the hand-over forms under uneven load (in a child process on the test-hooks library), the cold-start checks'
oracle-side twin, the frame call's instrumentation."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ape_oracle as orc

pytestmark = pytest.mark.gpu


def test_hand_overs_under_uneven_load():
    """tests/hooks/uneven_load_cases.py: every flag-based kernel (lstm_cluster32 both instantiations, lstm_cluster, lstm_cluster16,
    lstm_cluster_f16v2, lstm_cluster_f16, mlp_pipe, lstm_upper32, lstm_upper128) and the two tagged-granule latency kernels beside a queue
    of large device copies on a second stream, every output row against the oracle.  One child process, one pytest run."""
    from tests.conftest import REPO
    lib = REPO / "arm-pose-estimation_amd" / "lib" / "diag" / "libape_hip_testhooks.so"
    assert lib.exists(), "make -C arm-pose-estimation_amd/csrc hooks"
    torch.cuda.synchronize()
    env = dict(os.environ, APE_HIP_LIB=str(lib))
    r = subprocess.run([sys.executable, "-m", "pytest", str(REPO / "tests" / "hooks" / "uneven_load_cases.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=env, cwd=str(REPO), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-400:]
