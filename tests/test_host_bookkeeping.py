"""CPU tests of the host side: index/column bookkeeping (bit-exact vs the reference's tables),
the C-ABI library (loads, exports every declared symbol; no compute without a GPU), the
loader's error behaviour and the host feature builder vs golden rows."""
import ctypes as C
import json
import re
from array import array
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
GOLDEN = REPO / "tests" / "golden"
BOOK = json.loads((GOLDEN / "bookkeeping.json").read_text())


# ---------------- names / lookups: bit-exact ---------------------------------------------------
def test_column_enums_match_reference():
    from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS, TARGET_LAYOUT
    assert {m.name: m.value for m in NNS_INPUTS} == BOOK["NNS_INPUTS"]
    assert {m.name: m.value for m in NNS_TARGETS} == BOOK["NNS_TARGETS"]
    assert [len(NNS_INPUTS[n].value) for n in ("WATCH_ONLY_CAL", "WATCH_PHONE_CAL_HIP", "WATCH_PHONE_CAL_ALL")] == [20, 22, 38]
    assert {t.name: len(t.value) for t in TARGET_LAYOUT} == {"ORI_CAL_LARM_UARM_HIPS": 14, "ORI_CAL_LARM_UARM": 12,
                                                             "ORI_POS_CAL_LARM_UARM_HIPS": 20}
    # lookup by name, as the estimators do with results.json values
    assert NNS_INPUTS["WATCH_PHONE_CAL_HIP"].name == "WATCH_PHONE_CAL_HIP"


def test_message_lookups_match_reference():
    from wear_mocap_ape_amd.data_types import messaging
    assert messaging.WATCH_ONLY_IMU_LOOKUP == BOOK["WATCH_ONLY_IMU_LOOKUP"]
    assert messaging.WATCH_PHONE_IMU_LOOKUP == BOOK["WATCH_PHONE_IMU_LOOKUP"]
    assert messaging.watch_only_imu_msg_len == BOOK["watch_only_imu_msg_len"] == 112
    assert messaging.watch_phone_imu_msg_len == BOOK["watch_phone_imu_msg_len"] == 220


def test_deploy_registry_and_body_defaults():
    from wear_mocap_ape_amd.data_deploy.nn import deploy_models
    from wear_mocap_ape_amd.data_types.bone_map import BoneMap
    assert {m.name: m.value for m in deploy_models.LSTM} == BOOK["deploy_hashes"]
    assert BoneMap.DEFAULT_LARM_LEN == BOOK["bone_defaults"]["larm"]
    assert BoneMap.DEFAULT_UARM_LEN == BOOK["bone_defaults"]["uarm"]
    assert BoneMap.DEFAULT_UARM_ORIG_RH.tolist() == BOOK["bone_defaults"]["uarm_orig_rh"]


def test_norm_stats_load_branch(norm_stats):
    from wear_mocap_ape_amd.utility import data_stats
    from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS
    st = data_stats.get_norm_stats(NNS_INPUTS.WATCH_PHONE_CAL_HIP, NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS)
    for k in ("xx_m", "xx_s", "yy_m", "yy_s"):
        assert st[k].dtype == np.float64
        assert np.array_equal(st[k], norm_stats["pocket"][k])
    with pytest.raises(UserWarning):   # no stats file for this pair and no data list (data_stats.py:50-51)
        data_stats.get_norm_stats(NNS_INPUTS.WATCH_ONLY_RAW, NNS_TARGETS.ORI_CAL_LARM_UARM)


# ---------------- C ABI: library loads and exports what the header declares ----------------------
def _declared_functions():
    text = (REPO / "include" / "ape_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ape_[a-z0-9_]+)\s*\(", text)))


def test_c_abi_exports_every_declared_symbol():
    import __graft_entry__ as entry
    entry.build()                                # hipcc cross-compiles gfx950 without a GPU
    from wear_mocap_ape_amd import _hip
    declared = _declared_functions()
    assert declared, "no prototypes found in include/ape_hip.h"
    raw = C.CDLL(str(_hip.LIB_PATH))
    for name in declared:
        assert hasattr(raw, name), f"libape_hip.so does not export {name}"
    assert sorted(_hip.SIGNATURES) == declared   # the Python binding covers the same set
    lib = _hip.lib()
    assert lib.ape_abi_version() == _hip.ABI_VERSION == 7
    assert not hasattr(raw, "ape_debug_poke"), "test hooks belong to lib/diag/libape_hip_testhooks.so only"
    assert lib.ape_device_count() >= 0


def test_c_abi_argument_checks_without_gpu():
    from wear_mocap_ape_amd import _hip
    lib = _hip.lib()
    bad = _hip.ApeDims(22, 200, 2, 14, 0, 0)            # hidden size the kernels are not built for
    assert lib.ape_weight_blob_floats(C.byref(bad)) == 0
    assert b"hidden_size" in lib.ape_last_error()
    pocket = _hip.ApeDims(22, 256, 2, 14, 0, 0)
    assert lib.ape_weight_blob_floats(C.byref(pocket)) == 816654
    assert lib.ape_weight_blob_floats(C.byref(_hip.ApeDims(20, 256, 2, 12, 1, 0))) == 814092
    assert lib.ape_weight_blob_floats(C.byref(_hip.ApeDims(38, 128, 3, 12, 1, 0))) == 351756
    assert lib.ape_flops_per_window(C.byref(pocket), 64) == 103554048     # SURVEY.md 8d (incl. the head)
    assert lib.ape_flops_per_window(C.byref(pocket), 6) == 9714688
    mismatch = _hip.ApeDims(22, 256, 2, 12, 0, 0)       # layout 0 needs 14 targets
    assert lib.ape_weight_blob_floats(C.byref(mismatch)) == 0
    if lib.ape_device_count() == 0:
        h = C.c_void_p()
        rc = lib.ape_model_create(C.byref(pocket), C.byref(h))
        assert rc != 0 and not h.value                    # loud failure, no CPU fallback
        with pytest.raises(UserWarning):
            _hip.check(rc, "ape_model_create")


def test_loader_errors_are_userwarnings(tmp_path, monkeypatch):
    from wear_mocap_ape_amd import config
    from wear_mocap_ape_amd.estimate import nn_models
    monkeypatch.setitem(config.PATHS, "deploy", tmp_path)
    with pytest.raises(UserWarning, match="no json found"):
        nn_models.load_deployed_model_from_hash("deadbeef")
    d = tmp_path / "nn" / "deadbeef"
    d.mkdir(parents=True)
    (d / "results.json").write_text(json.dumps({"model": "DropoutLSTM"}))
    with pytest.raises(UserWarning, match="no checkpoint found"):
        nn_models.load_deployed_model_from_hash("deadbeef")
    (d / "checkpoint.pt").write_bytes(b"")
    (d / "results.json").write_text(json.dumps({"model": "OneHotLSTM"}))       # not reachable from the loader upstream either
    with pytest.raises(UserWarning, match="not handled"):
        nn_models.load_deployed_model_from_hash("deadbeef")


def test_shipped_results_json_have_loader_fields():
    from wear_mocap_ape_amd import config
    from wear_mocap_ape_amd.data_deploy.nn import deploy_models
    dims = {}
    for m in deploy_models.LSTM:
        p = json.loads((Path(config.PATHS["deploy"]) / "nn" / m.value / "results.json").read_text())
        dims[m.name] = (len(p["x_inputs_v"]), p["hidden_layer_size"], p["hidden_layer_count"], len(p["y_targets_v"]),
                        p["sequence_len"], p["model"], p["normalize"])
    assert dims == {"WATCH_PHONE_POCKET": (22, 256, 2, 14, 6, "DropoutLSTM", True),
                    "WATCH_PHONE_UARM": (38, 128, 3, 12, 6, "DropoutLSTM", True),
                    "WATCH_ONLY": (20, 256, 2, 12, 8, "DropoutLSTM", True)}


# ---------------- window bookkeeping (host list logic, no GPU) -----------------------------------
def test_push_padded_matches_reference_loops():
    from wear_mocap_ape_amd.estimate.estimator import Estimator

    def reference_loops(hist, item, size):       # estimator.py:94-100 restated literally
        hist.append(item)
        while len(hist) < size:
            hist.append(item)
        while len(hist) > size:
            del hist[0]

    for size in (1, 3, 6):
        a, b = [], []
        for item in range(12):
            Estimator._push_padded(a, item, size)
            reference_loops(b, item, size)
            assert a == b and len(a) == size
    # a history longer than the window (e.g. after seq_len shrank) is trimmed from the front
    a = list(range(10))
    Estimator._push_padded(a, 99, 4)
    assert a == [8, 9, 99][-4:] or a == [7, 8, 9, 99]


# ---------------- feature builder (host, f1 "next" row) vs golden rows ------------------------------
@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_features_from_row(golden, name):
    from wear_mocap_ape_amd.data_types import messaging
    import wear_mocap_ape_amd.estimate.watch_only as wo
    import wear_mocap_ape_amd.estimate.watch_phone_pocket_nn as wp
    import wear_mocap_ape_amd.estimate.watch_phone_uarm_nn as wu
    g = golden(f"stream_trace_{name}.npz")
    fn, lookup = {"pocket": (wp.features_from_row, messaging.WATCH_PHONE_IMU_LOOKUP),
                  "watch": (wo.features_from_row, messaging.WATCH_ONLY_IMU_LOOKUP),
                  "uarm": (wu.features_from_row, messaging.WATCH_PHONE_IMU_LOOKUP)}[name]
    ref = g["xx_s1_mc1"]
    dtype = str(g["xx_dtype_s1_mc1"])
    # uarm: the reference's np.array(array('f')) makes part of its quaternion math float32
    tol = 2e-6 if name == "uarm" else (1e-6 if dtype == "float32" else 1e-12)
    for f, row32 in enumerate(g["rows"]):
        xx = fn(array("f", row32.tolist()), lookup)
        assert str(xx.dtype) == dtype and xx.shape == ref[f].shape
        assert np.abs(xx.astype(np.float64) - ref[f]).max() < tol, (f, np.abs(xx - ref[f]).max())


@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_features_from_row_over_the_azimuths(golden, name):
    """the same builder on `feature_edges.npz`: 74 rows whose calibration quaternions sweep the azimuth circle, its corners (0, +-pi, +-pi/2,
    pi - 1e-6, +-1e-7), tilted and un-normalised poses (1e-18 .. 1e15) and no calibration at all (the zero quaternion: NaN where the
    reference has NaN) -- the reference's own parse_row_to_xx outputs (tests/golden/gen_golden.py gen_feature_edges)"""
    from wear_mocap_ape_amd.data_types import messaging
    import wear_mocap_ape_amd.estimate.watch_only as wo
    import wear_mocap_ape_amd.estimate.watch_phone_pocket_nn as wp
    import wear_mocap_ape_amd.estimate.watch_phone_uarm_nn as wu
    g = golden("feature_edges.npz")
    fn, lookup = {"pocket": (wp.features_from_row, messaging.WATCH_PHONE_IMU_LOOKUP),
                  "watch": (wo.features_from_row, messaging.WATCH_ONLY_IMU_LOOKUP),
                  "uarm": (wu.features_from_row, messaging.WATCH_PHONE_IMU_LOOKUP)}[name]
    rows, ref = g[f"rows_{name}"], g[f"xx_{name}"]
    assert len(rows) == 74 and np.isnan(ref).sum() == {"pocket": 2, "watch": 0, "uarm": 12}[name]
    with np.errstate(all="ignore"):
        xx = np.array([np.asarray(fn(array("f", r.tolist()), lookup), dtype=np.float64) for r in rows])
    assert np.array_equal(np.isnan(xx), np.isnan(ref))
    assert np.nanmax(np.abs(xx - ref)) < (1e-12 if name == "uarm" else 1e-7), np.nanmax(np.abs(xx - ref))      # measured: 6.7e-16 / 0.0


def build_c_caller(tmp_path):
    """gcc-compile tests/c_abi/demo.c (plain C, no Python, no torch) against include/ape_hip.h and libape_hip.so"""
    import subprocess
    import __graft_entry__ as entry
    entry.build()
    exe = tmp_path / "c_abi_demo"
    lib_dir = REPO / "arm-pose-estimation_amd" / "lib"
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{REPO / 'include'}",
           str(REPO / "tests" / "c_abi" / "demo.c"), f"-L{lib_dir}", "-lape_hip", "-L/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_caller_compiles_and_links(tmp_path):
    """the header is valid C (not only C++) and every entry point the C caller uses resolves at link time"""
    exe = build_c_caller(tmp_path)
    assert exe.exists() and exe.stat().st_size > 0


def test_csv_recorder_header_and_rows(tmp_path):
    """the on-disk format of the pose messages -- header identical to the reference recorder's (fixture
    written by the reference, record/est_output.py:16-32), one line per queued message, queue protocol unchanged"""
    import json, queue, time
    from wear_mocap_ape_amd.record.est_output import EstOutputRecorder, msg_columns
    want = json.loads((GOLDEN / "est_csv_header.json").read_text())
    assert ["time"] + msg_columns() == want and len(msg_columns()) == 25
    with pytest.raises(UserWarning):
        EstOutputRecorder(tmp_path / "missing_dir" / "est.csv")
    f = tmp_path / "est.csv"
    rec = EstOutputRecorder(f)
    assert f.read_text().strip().split(",") == want
    q = queue.Queue()
    rec.record_in_thread(q)
    msgs = [np.arange(25, dtype=np.float64) + 0.5 * i for i in range(3)]
    for m in msgs:
        q.put(m)
    deadline = time.time() + 10
    while len(f.read_text().strip().splitlines()) < 4 and time.time() < deadline:
        time.sleep(0.05)
    rec.terminate()
    lines = f.read_text().strip().splitlines()
    assert len(lines) == 4
    for line, m in zip(lines[1:], msgs):
        cells = line.split(",")
        assert len(cells) == 26 and [float(c) for c in cells[1:]] == list(m)
