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


# tests/hooks/uneven_load_cases.py runs ONCE, in a child process on the test-hooks library (the product objects + the injected-mask hook the bank
# routes need); each of its cases is a test of THIS collection, so that a red case has its own name in the driver's record (VERDICT r05
# item 4c: the whole file used to count as one test, a failure visible only in a 4000-character tail).  The list is static -- collection
# must not need a GPU or a subprocess -- and the fixture fails every case if the child ran a different set.
UNEVEN_LOAD_CASES = [
    "test_lstm_kernels_under_uneven_load[pocket-640-16-f32-cluster-ape_lstm_cluster32]",
    "test_lstm_kernels_under_uneven_load[pocket-1024-64-f32-cluster-ape_lstm_cluster32]",
    "test_lstm_kernels_under_uneven_load[pocket-1024-6-f32-cluster-ape_lstm_cluster32]",
    "test_lstm_kernels_under_uneven_load[watch-600-8-f32-cluster-ape_lstm_cluster32]",
    "test_lstm_kernels_under_uneven_load[pocket-512-8-f32-cluster_gen1-ape_lstm_cluster]",
    "test_lstm_kernels_under_uneven_load[pocket-300-6-f32-cluster_gen1-ape_lstm_cluster]",
    "test_lstm_kernels_under_uneven_load[uarm-1024-6-f32-cluster_gen1-ape_lstm_cluster]",
    "test_lstm_kernels_under_uneven_load[uarm-700-50-f32-cluster-ape_lstm_cluster16]",
    "test_lstm_kernels_under_uneven_load[uarm-1024-64-f32-cluster-ape_lstm_cluster16]",
    "test_lstm_kernels_under_uneven_load[uarm-1024-6-f32-auto-ape_lstm_level16]",
    "test_lstm_kernels_under_uneven_load[uarm-700-13-f32-auto-ape_lstm_level16]",
    "test_lstm_kernels_under_uneven_load[uarm-530-2-f32-auto-ape_lstm_level16]",
    "test_lstm_kernels_under_uneven_load[uarm-300-6-f32-auto-ape_lstm_level16]",
    "test_lstm_kernels_under_uneven_load[watch-700-8-f16-cluster-ape_lstm_cluster_f16v2]",
    "test_lstm_kernels_under_uneven_load[watch-1024-64-f16-cluster-ape_lstm_cluster_f16v2]",
    "test_lstm_kernels_under_uneven_load[pocket-200-6-f16_gen1-cluster-ape_lstm_cluster_f16]",
    "test_lstm_kernels_under_uneven_load[pocket-1-6-f32-auto-ape_lstm_cluster_small]",
    "test_lstm_kernels_under_uneven_load[pocket-4-6-f32-auto-ape_lstm_cluster_small]",
    "test_mlp_pipeline_under_uneven_load",
    "test_bank_routes_under_uneven_load[pocket-170-25-ape_lstm_upper32]",
    "test_bank_routes_under_uneven_load[watch-100-25-ape_lstm_upper32]",
    "test_bank_routes_under_uneven_load[pocket-30-25-ape_lstm_upper32]",
    "test_bank_routes_under_uneven_load[watch-50-25-ape_lstm_upper32]",
    "test_bank_routes_under_uneven_load[uarm-100-50-ape_lstm_upper128]",
    "test_bank_routes_under_uneven_load[uarm-160-25-ape_lstm_upper128]",
    "test_bank_routes_under_uneven_load[uarm-30-50-ape_lstm_upper128]",
    "test_bank_routes_under_uneven_load[pocket-1-25-ape_lstm_mc_small]",
    "test_fresh_banks_layer0_sequence_under_uneven_load",
    "test_imupose_layer_split_under_uneven_load[1024-9]",
    "test_imupose_layer_split_under_uneven_load[1500-5]",
    "test_imupose_layer_split_under_uneven_load[2090-4]",
]


@pytest.fixture(scope="module")
def uneven_load_results(tmp_path_factory):
    """one child pytest over the whole file (no -x: every case reports), its junit record parsed into {case id: (outcome, text)}"""
    import xml.etree.ElementTree as ET
    from tests.conftest import REPO
    lib = REPO / "arm-pose-estimation_amd" / "lib" / "diag" / "libape_hip_testhooks.so"
    assert lib.exists(), "make -C arm-pose-estimation_amd/csrc hooks"
    torch.cuda.synchronize()
    xml = tmp_path_factory.mktemp("uneven") / "cases.xml"
    env = dict(os.environ, APE_HIP_LIB=str(lib))
    r = subprocess.run([sys.executable, "-m", "pytest", str(REPO / "tests" / "hooks" / "uneven_load_cases.py"), "-q", "-m", "gpu",
                        "-p", "no:cacheprovider", f"--junitxml={xml}"], env=env, cwd=str(REPO), capture_output=True, text=True, timeout=1200)
    out = {}
    if xml.exists():
        for tc in ET.parse(str(xml)).getroot().iter("testcase"):
            bad = [c for c in tc if c.tag in ("failure", "error", "skipped")]
            out[tc.get("name")] = ("passed", "") if not bad else (bad[0].tag, (bad[0].get("message") or "") + "\n" + (bad[0].text or "")[-1500:])
    return {"cases": out, "rc": r.returncode, "tail": r.stdout[-1500:] + r.stderr[-500:]}


@pytest.mark.parametrize("case", UNEVEN_LOAD_CASES)
def test_hand_overs_under_uneven_load(uneven_load_results, case):
    """tests/hooks/uneven_load_cases.py: every flag-based kernel (lstm_cluster32 both instantiations, lstm_cluster, lstm_cluster16,
    lstm_cluster_f16v2, lstm_cluster_f16, mlp_pipe, lstm_upper32, lstm_upper128) and the tagged-granule kernels (the two latency kernels,
    lstm_level16) beside a queue of large device copies on a second stream, every output row against the oracle.
    (The build-time scan, tools/check_mfma_hazards.py, is the guard against the inline-asm hazard classes; of these cases the fresh-bank one
    would catch a regression of the store-data class in about one run of eight, DESIGN.md 4.18 -- they test the hand-over protocols.)"""
    res = uneven_load_results
    assert set(res["cases"]) == set(UNEVEN_LOAD_CASES), (sorted(set(res["cases"]) ^ set(UNEVEN_LOAD_CASES)), res["tail"])
    outcome, text = res["cases"][case]
    assert outcome == "passed", f"{case}: {outcome}\n{text}"
