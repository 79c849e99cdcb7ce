"""GPU parity tests added in round 4 (all through the C ABI of libape_hip.so):
  * the drop-in `Estimator` consumer loop on its device-resident frame (`ape_streams_frame_host`): the reference's own
    20-frame traces in eval mode, the reference estimators' Monte-Carlo distribution end to end.
"""
import queue
import time
from array import array

import numpy as np
import pytest
import torch

from oracle import ape_oracle as orc
from tests import mc_check
from tests.test_hip_parity import _deploy_dir

pytestmark = pytest.mark.gpu

TOL_MSG_LOOP = 5e-6          # float32 regressor in front of the float64 post-filter (5e-5 budget on quaternions / origins)


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as entry
    entry.build()


def _estimator_class(name):
    from wear_mocap_ape_amd.estimate.watch_only import WatchOnlyNN
    from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    from wear_mocap_ape_amd.estimate.watch_phone_uarm_nn import WatchPhoneUarmNN
    return {"pocket": WatchPhonePocketNN, "watch": WatchOnlyNN, "uarm": WatchPhoneUarmNN}[name]


@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_consumer_loop_on_the_device_frame_replays_reference_traces(golden, tmp_path, monkeypatch, name):
    """one `process_row` per raw message = one iteration of estimator.py:174-177 (parse_row_to_xx ->
    add_xx_to_row_hist_and_make_prediction -> msg_from_pred), with the histories on the device: the message lists of the
    reference's own 20-frame traces (eval-mode goldens: dropout 0), cold start, smoothing stack and the 25 + 6N tail included;
    then the same through the consumer thread, after a reset, and against the staged methods."""
    from wear_mocap_ape_amd import config
    g = golden(f"stream_trace_{name}.npz")
    deploy, h = _deploy_dir(tmp_path, name, int(g["weights_seed"]), dropout=0.0)
    monkeypatch.setitem(config.PATHS, "deploy", deploy)
    cls = _estimator_class(name)
    for smooth, mc in ((1, 1), (5, 1), (3, 4)):
        tag = f"s{smooth}_mc{mc}"
        est = cls(model_hash=h, smooth=smooth, add_mc_samples=True, monte_carlo_samples=mc)
        assert est._frame_runner() is not None, "the device-resident frame must be the path that runs"
        for rep in range(2):                         # second pass behind a reset: the same cold start
            worst = 0.0
            for f, row32 in enumerate(g["rows"]):
                msg = est.process_row(array("f", row32.tolist()))
                msg_ref = g[f"msg_{tag}"][f]
                n_rows = smooth * mc
                assert isinstance(msg, list) and len(msg) == len(msg_ref) == (25 + 6 * n_rows if n_rows > 1 else 25)   # bit-exact bookkeeping
                worst = max(worst, float(np.abs(np.asarray(msg) - msg_ref).max()))
                assert msg[0:4] == msg[7:11]                                         # compose_msg.py:72,74: hand rot = lower-arm rot
            assert worst < TOL_MSG_LOOP, (tag, rep, worst)
            last = est.get_last_msg()
            assert last.shape == (25,) and np.abs(last - g[f"last_msg_{tag}"]).max() < TOL_MSG_LOOP
            est.reset()
            assert est._row_hist == [] and est._smooth_hist == [] and not est.is_active()
        # add_mc_samples False: the 25-value array (estimator.py:129-130)
        est2 = cls(model_hash=h, smooth=smooth, add_mc_samples=False, monte_carlo_samples=mc)
        out = est2.process_row(array("f", g["rows"][0].tolist()))
        assert isinstance(out, np.ndarray) and out.shape == (25,) and np.abs(out - g[f"msg_{tag}"][0][:25]).max() < TOL_MSG_LOOP
        # the staged methods (reference semantics, host histories) stay available and agree
        est3 = cls(model_hash=h, smooth=smooth, add_mc_samples=True, monte_carlo_samples=mc)
        est3.use_device_frame = False
        assert est3._frame_runner() is None
        est.reset()
        for row32 in g["rows"][:8]:
            a = est.process_row(array("f", row32.tolist()))
            b = est3.process_row(array("f", row32.tolist()))
            assert len(a) == len(b) and np.abs(np.asarray(a) - np.asarray(b)).max() < TOL_MSG_LOOP
        assert est._hip_model().stats()["aborted_checks"] == 0
    # the consumer thread itself (estimator.py:139-178): sensor queue in, message queue out
    est = cls(model_hash=h, smooth=3, add_mc_samples=True, monte_carlo_samples=4)
    sensor_q = queue.Queue()
    msg_q = est.process_in_thread(sensor_q)
    try:
        for f, row32 in enumerate(g["rows"]):
            sensor_q.put(array("f", row32.tolist()))
            msg = msg_q.get(timeout=30)
            assert np.abs(np.asarray(msg) - g["msg_s3_mc4"][f]).max() < TOL_MSG_LOOP
        assert est.is_active()
    finally:
        est.terminate()
        time.sleep(0.1)


@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_consumer_loop_monte_carlo_matches_reference_estimators(golden, tmp_path, monkeypatch, name):
    """the estimators' deployed mode end to end: the reference classes themselves (deployed dropout rate, their own
    processing order) were driven through the 20-row trace and the hand / elbow rows of their LAST frame kept (24 000 samples,
    tests/golden/trace_mc_stats.npz).  Here the drop-in loop runs the same trace 128 times (reset in between, like
    processing_loop on start) with 64 samples per frame: the 8192 tail rows of the last frames must be samples of that
    distribution (tests/mc_check.py), the mean-pose message close to the reference's 4000-sample ones; a wrong rate is flagged."""
    from wear_mocap_ape_amd import config
    g = golden("trace_mc_stats.npz")
    rows = golden(f"stream_trace_{name}.npz")["rows"]
    p, n_ref, levels = float(g[f"dropout_{name}"]), int(g["n_samples"]), g["quantile_levels"]
    stats = (g[f"est6_mean_{name}"], g[f"est6_cov_{name}"], g[f"est6_quant_{name}"], levels, n_ref)
    cls = _estimator_class(name)
    wire = [array("f", r.tolist()) for r in rows]

    def last_frame_tails(dropout, passes, n_mc=64):
        deploy, h = _deploy_dir(tmp_path / f"p{dropout}", name, int(g["weights_seed"]), dropout=dropout)
        monkeypatch.setitem(config.PATHS, "deploy", deploy)
        est = cls(model_hash=h, smooth=1, add_mc_samples=True, monte_carlo_samples=n_mc)
        assert est._frame_runner() is not None
        tails, msgs = [], []
        for _ in range(passes):
            est.reset()
            for row in wire:
                msg = est.process_row(row)
            assert len(msg) == 25 + 6 * n_mc
            tails.append(np.asarray(msg[25:]).reshape(n_mc, 6))
            msgs.append(np.asarray(msg[:25]))
        assert est._hip_model().stats()["aborted_checks"] == 0
        return np.concatenate(tails), np.array(msgs)

    tails, msgs = last_frame_tails(p, 128)
    bad = mc_check.compare(tails, *stats, what=f"{name} consumer loop")
    assert not bad, bad
    # the message itself: sign-aligned quaternion means and origins of 64 samples scatter around the reference's 4000-sample
    # means by ~ sigma / 8; their average over the 128 passes must sit within a few standard errors
    ref_msg = g[f"msg_mean_{name}"].mean(axis=0)
    spread = msgs.std(axis=0) / np.sqrt(len(msgs)) + 1e-4
    assert np.all(np.abs(msgs.mean(axis=0) - ref_msg) < 6.0 * spread + 2e-3), np.abs(msgs.mean(axis=0) - ref_msg).max()
    # negative control: half the rate must be flagged
    tails_bad, _ = last_frame_tails(0.5 * p, 128)
    assert mc_check.compare(tails_bad, *stats)
