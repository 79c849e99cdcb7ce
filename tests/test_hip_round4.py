"""GPU parity tests added in round 4 (all through the C ABI of libape_hip.so):
  * the drop-in `Estimator` consumer loop on its device-resident frame (`ape_streams_frame_host`): the reference's own
    20-frame traces in eval mode, the reference estimators' Monte-Carlo distribution end to end.
"""
import ctypes as C
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
        # opt-in (round 6): the same message as ONE float array instead of a list -- what the reference's publisher packs either way
        # (struct.pack('f' * len(msg), *msg), stream/publisher/pose_est_udp.py:47): same values, same length, same bytes on the wire
        import struct
        est4 = cls(model_hash=h, smooth=smooth, add_mc_samples=True, monte_carlo_samples=mc)
        est4.msg_as_array = True
        est.reset()
        for f, row32 in enumerate(g["rows"][:6]):
            a = est.process_row(array("f", row32.tolist()))
            b = est4.process_row(array("f", row32.tolist()))
            assert isinstance(b, np.ndarray) and b.ndim == 1 and len(b) == len(a) and np.array_equal(np.asarray(a, dtype=b.dtype), b)
            assert struct.pack("f" * len(a), *a) == struct.pack("f" * len(b), *b)
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


# ---------------- the Monte-Carlo latency kernel (lstm_mc_small.hip) ---------------------------------------------------------
@pytest.mark.parametrize("name,n,T", [("pocket", 25, 6), ("pocket", 1, 6), ("pocket", 60, 6), ("pocket", 128, 3), ("pocket", 7, 1),
                                      ("pocket", 33, 13), ("watch", 25, 8), ("watch", 100, 2), ("uarm", 50, 6), ("uarm", 3, 6),
                                      ("uarm", 128, 9), ("uarm", 17, 1)])
def test_mc_latency_kernel_with_injected_masks(norm_stats, name, n, T):
    """one window, n dropout samples (monte_carlo_predictions, nn_models.py:191-207) on the Monte-Carlo latency kernel with the
    caller's masks, against the oracle's masked cell loop fed the same masks (1e-6 per sample row): row counts around the
    cluster capacities (4 / 8 / 16 rows on each of 8 XCDs), window lengths around the pipeline depth, all three deployed shapes;
    the fused z-score; the forced any-placement exchange; run-to-run determinism."""
    from tests.test_hip_parity import make_model
    from wear_mocap_ape_amd import _hip
    m, sd, cfg = make_model(name, 1, norm_stats[name])
    rng = np.random.default_rng(n * 100 + T)
    x = rng.normal(size=(1, T, cfg["I"])).astype(np.float32)
    p = 0.2
    masks = [(rng.random((n, T, cfg["H"])) >= p).astype(np.float32) / np.float32(1.0 - p) for _ in range(cfg["L"] - 1)]
    ref = orc.lstm_forward(sd, np.repeat(x, n, axis=0), masks=masks)[:, -1, :]
    md = torch.from_numpy(np.stack(masks)).cuda()
    y = m(torch.from_numpy(x).cuda(), masks=md, last_step_only=True, rows=n)
    assert m.last_kernel() == "ape_lstm_mc_small"
    y = y.cpu().numpy()[:, 0]
    assert y.shape == ref.shape and np.abs(y - ref).max() < 1e-6, np.abs(y - ref).max()
    y2 = m(torch.from_numpy(x).cuda(), masks=md, last_step_only=True, rows=n).cpu().numpy()[:, 0]
    assert np.array_equal(y, y2)
    # raw features + fused float64 z-score
    st = norm_stats[name]
    raw = (x.astype(np.float64) * st["xx_s"] + st["xx_m"]).astype(np.float32)
    xn = ((raw.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
    ref_n = orc.lstm_forward(sd, np.repeat(xn, n, axis=0), masks=masks)[:, -1, :]
    yn = m(torch.from_numpy(raw).cuda(), masks=md, last_step_only=True, rows=n, normalize_input=True).cpu().numpy()[:, 0]
    assert np.abs(yn - ref_n).max() < 1e-6
    # the other kernels under the same masks (other float32 summation orders only)
    m.set_kernel("tile16")
    yt = m(torch.from_numpy(x).cuda(), masks=md, last_step_only=True, rows=n).cpu().numpy()[:, 0]
    assert m.last_kernel() == "ape_lstm_tile16"
    m.set_kernel("auto")
    assert np.abs(y - yt).max() < 1e-6
    m.check()


@pytest.mark.parametrize("name,n", [("pocket", 25), ("pocket", 128), ("uarm", 50), ("watch", 60)])
def test_mc_latency_kernel_draws_the_masks_of_the_other_kernels(norm_stats, name, n):
    """in-kernel Philox: counters (row quad, step, unit, layer) and key as in every other kernel of the library, so the same seed
    gives the same samples whatever kernel the dispatch picks (batch-tile kernel: 1e-6 per row)"""
    from tests.test_hip_parity import make_model
    m, sd, cfg = make_model(name, 2, norm_stats[name])
    x = torch.from_numpy(np.random.default_rng(3).normal(size=(1, cfg["T"], cfg["I"])).astype(np.float32)).cuda()
    outs = {}
    for kernel in ("auto", "tile16", "auto_gen1"):
        m.set_kernel(kernel)
        m.manual_seed(77)
        outs[kernel] = m.monte_carlo_predictions(n, x, last_step_only=True).cpu().numpy()[:, 0]
        outs[kernel + "_name"] = m.last_kernel()
    m.set_kernel("auto")
    assert outs["auto_name"] == "ape_lstm_mc_small" and outs["tile16_name"] == "ape_lstm_tile16" and outs["auto_gen1_name"] == "ape_lstm_cluster"
    assert np.abs(outs["auto"] - outs["tile16"]).max() < 1e-6 and np.abs(outs["auto"] - outs["auto_gen1"]).max() < 1e-6
    assert np.abs(outs["auto"] - outs["auto"][0]).max() > 1e-3          # the samples differ from each other
    m.check()


@pytest.mark.parametrize("name", ["pocket", "uarm"])
def test_mc_latency_kernel_matches_reference_distribution(golden, name):
    """64 calls of `monte_carlo_predictions(128, x)` on the latency kernel = 8192 samples of one window, against the statistics of
    the 24 000 samples the REFERENCE drew (tests/golden/mc_stats.npz); half the rate is flagged"""
    from tests.test_hip_parity import make_model
    g = golden("mc_stats.npz")
    cfg = orc.MODEL_CONFIGS[name]
    m, sd, _ = make_model(name, 0)
    p, n_ref, levels = float(g[f"dropout_{name}"]), int(g["n_samples"]), g["quantile_levels"]
    m.manual_seed(31)
    for w in (0, 2):
        x = torch.from_numpy(g[f"x_{name}"][w:w + 1]).cuda()
        args = (g[f"y_mean_{name}"][w], g[f"y_cov_{name}"][w], g[f"y_quant_{name}"][w], levels, n_ref)
        ys = [m.monte_carlo_predictions(128, x, last_step_only=True)[:, 0].cpu().numpy() for _ in range(64)]
        assert m.last_kernel() == "ape_lstm_mc_small"
        bad = mc_check.compare(np.concatenate(ys), *args, what=f"{name} w{w} latency kernel")
        assert not bad, bad
    m.dropout = 0.5 * p
    x = torch.from_numpy(g[f"x_{name}"][0:1]).cuda()
    ys = [m.monte_carlo_predictions(128, x, last_step_only=True)[:, 0].cpu().numpy() for _ in range(64)]
    assert mc_check.compare(np.concatenate(ys), g[f"y_mean_{name}"][0], g[f"y_cov_{name}"][0], g[f"y_quant_{name}"][0], levels, n_ref)
    m.check()


@pytest.mark.parametrize("name,S,n_mc,smooth", [("pocket", 1, 25, 1), ("pocket", 1, 60, 5), ("pocket", 2, 64, 1), ("pocket", 8, 16, 2),
                                                ("pocket", 3, 25, 1), ("uarm", 1, 50, 1), ("uarm", 5, 11, 1), ("watch", 1, 25, 10)])
def test_small_monte_carlo_banks_run_on_the_latency_kernel(norm_stats, name, S, n_mc, smooth):
    """a bank of up to eight streams in Monte-Carlo mode (one estimator's frame: S = 1) steps on the latency kernel: every sample
    against `ape_lstm_forward` on explicitly repeated windows with the bank's key (the batch-tile kernel under the same Philox
    counters, 5e-6 on hand / elbow positions and messages), window ring and smoothing stack against the oracle's bookkeeping"""
    from tests.test_hip_parity import make_model, _synthetic_windows
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    st = norm_stats[name]
    m, sd, cfg = make_model(name, 4, st)
    m.set_body(orc.DEFAULT_BODY)
    T, I, O = cfg["T"], cfg["I"], cfg["O"]
    frames = T + 3
    feats = _synthetic_windows(st, S, frames, I, 50 + S)
    seed, p = 1234, 0.2
    bank = StreamBank(m, S, T, smooth=smooth, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=p, seed=seed)
    hist = [[] for _ in range(S)]
    stack = [[] for _ in range(S)]
    for f in range(frames):
        bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda())
        msg, tail = bank.step(with_tail=True)
        assert m.last_kernel() == "ape_lstm_mc_small"
        msg, tail = msg.cpu().numpy().copy(), tail.cpu().numpy().copy()
        # the same samples through ape_lstm_forward on explicitly repeated windows (rows s * n_mc + k), the bank's key for this step
        wins = []
        for s in range(S):
            hist[s].append(feats[s, f])
            while len(hist[s]) < T:
                hist[s].append(feats[s, f])
            del hist[s][:len(hist[s]) - T]
            wins.append(np.repeat(np.stack(hist[s])[None], n_mc, axis=0))
        xw = torch.from_numpy(np.concatenate(wins).astype(np.float32)).cuda()
        y = torch.empty((S * n_mc, O), dtype=torch.float32, device="cuda")
        m.set_kernel("tile16")
        _hip.check(_hip.lib().ape_lstm_forward(m.handle, C.c_void_p(xw.data_ptr()), S * n_mc, T,
                                               _hip.FLAG_NORMALIZE_INPUT | _hip.FLAG_DROPOUT_PHILOX, None, p, seed + f,
                                               C.c_void_p(y.data_ptr()), None), "fwd")
        m.set_kernel("auto")
        yd = y.cpu().numpy().astype(np.float64) * st["yy_s"] + st["yy_m"]
        for s in range(S):
            pred = yd[s * n_mc:(s + 1) * n_mc]
            stack[s].append(pred)
            while len(stack[s]) < smooth:
                stack[s].append(pred)
            del stack[s][:len(stack[s]) - smooth]
            rows = np.vstack(stack[s])
            est = orc.arm_pose_from_targets(rows, orc.DEFAULT_BODY, cfg["layout"], "closed")
            ref_msg = orc.msg_from_est(est, orc.DEFAULT_BODY, cfg["layout"])
            assert np.abs(tail[s] - est[:, :6]).max() < 5e-6, (f, s)
            assert np.abs(msg[s] - ref_msg).max() < 5e-6, (f, s)
    m.check()


# ---------------- Philox counters name global rows in every kernel (round 3's unexplained wrong result) ------------------------------
@pytest.mark.parametrize("name,B,T", [("pocket", 2460, 6), ("pocket", 5000, 6), ("uarm", 1500, 6), ("watch", 700, 8)])
def test_philox_samples_do_not_depend_on_the_split(norm_stats, name, B, T):
    """gpurun_out/gpu_suite_r03e.log, `test_mc_bank_cluster_route_against_batch_tile_route[pocket-41-60-6]` off by 2.3e-2: with the
    dropout launch priced at what it measures, AUTO serves 2460 dropout rows with five first-generation cluster launches, and those drew
    their masks from a per-launch seed and launch-local row counters -- other samples than the routes that count global rows (the
    batch-tile kernel, the bank's weight-stationary route).  Now every kernel feeds Philox the GLOBAL row of the call: whatever the
    split (several cluster launches; batch-tile waves in front of a cluster rest), the samples are those of the batch-tile kernel."""
    from tests.test_hip_parity import make_model, _synthetic_windows
    from wear_mocap_ape_amd import _hip
    m, sd, cfg = make_model(name, 6, norm_stats[name])
    x = torch.from_numpy(_synthetic_windows(norm_stats[name], B, T, cfg["I"], 17)).cuda()
    lib = _hip.lib()
    outs = {}
    for kernel in ("auto", "tile16", "auto_gen1"):
        m.set_kernel(kernel)
        y = torch.empty((B, cfg["O"]), dtype=torch.float32, device="cuda")
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT | _hip.FLAG_DROPOUT_PHILOX,
                                        None, 0.2, 987654321, C.c_void_p(y.data_ptr()), None), "fwd")
        outs[kernel] = (y.cpu().numpy(), m.last_kernel())
    m.set_kernel("auto")
    m.check()
    assert outs["tile16"][1] == "ape_lstm_tile16" and outs["auto"][1] == "ape_lstm_cluster"      # (the rest of an AUTO split is a cluster launch)
    for kernel in ("auto", "auto_gen1"):
        d = np.abs(outs[kernel][0] - outs["tile16"][0]).max()
        assert d < 1e-6, (kernel, d)
    # the dropout is really on
    m2, _, _ = make_model(name, 6, norm_stats[name])
    assert np.abs(outs["tile16"][0] - m2(x, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]).max() > 1e-3


# ---------------- AUTO's dispatch thresholds on the box that runs the tests ----------------------------------------------------------
def test_auto_dispatch_names_the_planned_kernel_at_every_boundary():
    """bench.py's `dispatch_boundaries` leg runs AUTO and every forceable kernel on the same inputs at each threshold of the plan.  The
    deterministic part belongs here: the kernels the plan names at either side of each boundary (`ape_model_last_kernel`).  The timing
    ratios (AUTO over the fastest candidate) are the bench line's to report -- a wall-clock assertion inside the correctness suite flakes
    on a shared or throttled box and says nothing about correctness."""
    import bench
    d = bench.dispatch_boundaries(n_iter=2)
    assert "error" not in d, d
    k = {key: v["kernels"]["auto"] for key, v in d.items() if isinstance(v, dict)}
    assert k["pocket_eval_B512_T64"] == "ape_lstm_cluster" and k["pocket_eval_B513_T64"] == "ape_lstm_cluster32"
    assert k["uarm_eval_B1024_T48"] == "ape_lstm_level16" and k["uarm_eval_B1024_T49"] == "ape_lstm_cluster16"
    assert k["uarm_eval_B4_T6"] == "ape_lstm_cluster_small" and k["uarm_eval_B5_T6"] == "ape_lstm_level16"
    assert k["uarm_eval_B1024_T6"] == "ape_lstm_level16" and k["uarm_eval_B1025_T6"] == "ape_lstm_cluster"
    assert k["pocket_eval_B4_T6"] == "ape_lstm_cluster_small" and k["pocket_eval_B5_T6"] == "ape_lstm_cluster"
    assert k["pocket_mc_one_window_n128_T6"] == "ape_lstm_mc_small" and k["pocket_mc_one_window_n129_T6"] == "ape_lstm_cluster"
    assert k["pocket_mc_bank_512_sample_rows_T6"] == "ape_lstm_cluster" and k["pocket_mc_bank_513_sample_rows_T6"] == "ape_lstm_upper32"


def test_undeclared_flag_bits_are_refused(norm_stats):
    from tests.test_hip_parity import make_model
    from wear_mocap_ape_amd import _hip
    m, sd, cfg = make_model("pocket", 1, norm_stats["pocket"])
    x = torch.zeros((8, 6, cfg["I"]), device="cuda")
    y = torch.zeros((8, cfg["O"]), device="cuda")
    lib = _hip.lib()
    for bit in (0x40000000, 0x20000000, 0x10000000, 0x04000000, 0x00200000, 0x00100000, 0x100):
        assert lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), 8, 6, bit, None, 0.0, 0, C.c_void_p(y.data_ptr()), None) != 0, hex(bit)
    for bit in (_hip.FLAG_ANY_PLACEMENT, _hip.FLAG_IN_XCD_PLAIN, _hip.FLAG_NO_XCD_CLASSES, _hip.FLAG_ALT_FORM):
        assert lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), 8, 6, bit, None, 0.0, 0, C.c_void_p(y.data_ptr()), None) == 0, hex(bit)
    m.check()


@pytest.mark.parametrize("name,S,n_mc,route", [("pocket", 170, 25, "ape_lstm_upper32"), ("uarm", 100, 50, "ape_lstm_upper128")])
def test_bank_cooperative_routes_follow_a_later_set_kernel(norm_stats, name, S, n_mc, route):
    """round-3 advisor: the bank's weight-stationary routes are planned by ape_streams_set_mc, but `set_kernel('tile16')` -- the documented
    way to keep persistent clusters off a shared GPU -- must reach a bank that already exists: its next steps run on the batch-tile
    kernel (same Philox masks: within the float32 budget of the route before), and come back with `set_kernel('auto')`, bit-equal."""
    from tests.test_hip_parity import make_model, _synthetic_windows
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    st = norm_stats[name]
    m, sd, cfg = make_model(name, 6, st)
    m.set_body(orc.DEFAULT_BODY)
    T = cfg["T"]
    feats = _synthetic_windows(st, S, T + 2, cfg["I"], 77)

    def run(choices):
        bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=5)
        outs = []
        for f in range(T + 2):
            m.set_kernel(choices[f])
            bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda())
            msg, tail = bank.step(with_tail=True)
            assert m.last_kernel() == (route if choices[f] == "auto" else "ape_lstm_tile16"), (f, m.last_kernel())
            outs.append((msg.cpu().numpy().copy(), tail.cpu().numpy().copy()))
        m.set_kernel("auto")
        m.check()
        return outs

    ref = run(["auto"] * (T + 2))
    got = run(["auto"] * 3 + ["tile16", "auto_gen1"] + ["auto"] * (T - 3))
    for f, ((a, b), (c, d)) in enumerate(zip(ref, got)):
        if f in (3, 4):
            assert np.abs(a - c).max() < 5e-5 and np.abs(b - d).max() < 5e-5 and np.abs(a - c).max() > 0.0
        else:
            assert np.array_equal(a, c) and np.array_equal(b, d), f


# ---------------- the post-filter of banks without stacking: lanes = streams (round 4) --------------------------------------------------
@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_wide_post_kernel_equals_the_per_stream_one(norm_stats, name, dtype):
    """banks with smooth = 1 and one sample per stream run `ape_stream_post_wide_kernel` from 8 streams on (64 streams per workgroup);
    below that every stream has a workgroup of its own (`ape_stream_post_kernel`).  Same device functions in the same order: the first
    streams of a 70-stream bank (two workgroups of the wide form, the second ragged) must equal, bit for bit, a 7-stream bank fed the
    same rows -- message, 6-float tails and packed datagram rows -- and the oracle's FK on the bank's own targets (1e-11 in float64)."""
    from tests.test_hip_parity import make_model, _synthetic_windows
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    st = norm_stats[name]
    m, sd, cfg = make_model(name, 6, st)
    m.set_body(orc.DEFAULT_BODY)
    T, I = cfg["T"], cfg["I"]
    S_wide, S_narrow = 70, 7
    feats = _synthetic_windows(st, S_wide, T + 2, I, 77)
    wide = StreamBank(m, S_wide, T, smooth=1, normalize=True, dtype=dtype)
    narrow = StreamBank(m, S_narrow, T, smooth=1, normalize=True, dtype=dtype)
    for f in range(T + 2):
        fw = torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda()
        wide.push_features(fw)
        narrow.push_features(fw[:S_narrow].contiguous())
        if f % 2 == 0:
            mw, tw = (v.cpu().numpy().copy() for v in wide.step(with_tail=True))
            mn, tn = (v.cpu().numpy().copy() for v in narrow.step(with_tail=True))
            assert np.array_equal(mw[:S_narrow], mn) and np.array_equal(tw[:S_narrow], tn), f
            assert np.isfinite(mw).all() and np.isfinite(tw).all()
            # rows 64 .. 69 are the ragged second workgroup: the message's origins are the tail's
            assert np.array_equal(mw[:, 4:7], tw.reshape(S_wide, -1)[:, 0:3]) and np.array_equal(mw[:, 11:14], tw.reshape(S_wide, -1)[:, 3:6])
        else:
            # packed rows [25 + 6] straight through the C ABI (the Python surface, like the reference, adds a tail only to stacked rows)
            def packed(bank, S):
                out = torch.empty((S, 31), dtype=dtype, device="cuda")
                _hip.check(_hip.lib().ape_streams_step(bank._handle, bank._flags | _hip.FLAG_PACKED_MSG, C.c_void_p(out.data_ptr()), None,
                                                       bank._sel, None), "ape_streams_step")
                torch.cuda.synchronize()
                return out.cpu().numpy()
            dw, dn = packed(wide, S_wide), packed(narrow, S_narrow)
            assert np.array_equal(dw[:S_narrow], dn) and np.isfinite(dw).all(), f
            assert np.array_equal(dw[:, 4:7], dw[:, 25:28]) and np.array_equal(dw[:, 11:14], dw[:, 28:31])
    m.check()


@pytest.mark.parametrize("name,n_mc,smooth", [("pocket", 13, 6), ("uarm", 50, 3), ("watch", 25, 10)])
def test_split_post_kernel_equals_the_per_stream_one(norm_stats, name, n_mc, smooth):
    """stacks of more than 64 rows of a FEW streams are dealt over several workgroups (`ape_stream_post_split_kernel`: 64-row chunks,
    partial sums joined by the last workgroup to arrive, in chunk order); a bank too large for that (streams x chunks > CUs) keeps one
    workgroup per stream.  Same rows, same Philox rows (global row = stream * n_mc + sample) -> the first streams of a 140-stream bank
    against a 2-stream bank, both on the batch-tile kernel (row-independent: the same targets bit for bit): tails bit-equal (row-local
    arithmetic), messages to 1e-12 (the means' summation order differs), cold start and ring wrap-around included."""
    from tests.test_hip_parity import make_model, _synthetic_windows
    from wear_mocap_ape_amd.streams import StreamBank
    st = norm_stats[name]
    m, sd, cfg = make_model(name, 8, st)
    m.set_body(orc.DEFAULT_BODY)
    m.set_kernel("tile16")
    T, I = cfg["T"], cfg["I"]
    S_big, S_small = 140, 2
    feats = _synthetic_windows(st, S_big, smooth + 3, I, 91)
    big = StreamBank(m, S_big, T, smooth=smooth, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=77)
    small = StreamBank(m, S_small, T, smooth=smooth, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=77)
    for f in range(smooth + 3):
        fw = torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda()
        big.push_features(fw)
        small.push_features(fw[:S_small].contiguous())
        mb, tb = (v.cpu().numpy().copy() for v in big.step(with_tail=True))
        ms, ts = (v.cpu().numpy().copy() for v in small.step(with_tail=True))
        assert np.array_equal(tb[:S_small], ts), f
        assert np.abs(mb[:S_small] - ms).max() < 1e-12, (f, float(np.abs(mb[:S_small] - ms).max()))
    m.check()


# ---------------- cold starts: the first launch of a fresh process (round 4's stale-slice finding) -----------------------------------------
@pytest.mark.parametrize("args", [("cold_stress.py", "pocket", "1024", "6", "f32"), ("cold_stress.py", "pocket", "1024", "64", "f32"),
                                  ("cold_stress.py", "watch", "1024", "64", "f16"), ("cold_stress.py", "uarm", "1024", "64", "f32"),
                                  ("cold_bank.py", "uarm", "170", "50"), ("cold_bank.py", "pocket", "170", "25")])
def test_first_launch_of_a_fresh_process(args):
    """every flag-based cooperative kernel on the FIRST launch of a fresh process (cold clocks, cold caches, untouched exchange buffers)
    against the ORACLE and, beside it, the batch-tile kernel on the same input, in a child process: with plain hand-over stores
    `ape_lstm_upper128` read stale slices there in 7 of 8 runs (whole 32-row tiles off by 1e-3 .. 1e-2) while every warm launch was exact
    -- no test of this suite, all of them warm by the time they compare anything, could see it.  Round 5: the cold launch is compared with
    the oracle itself (the bank routes under injected masks, on the test-hooks library), so the check does not depend on a second HIP
    kernel being right on a cold chip."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if args[0] == "cold_bank.py":
        env["APE_HIP_LIB"] = os.path.join(root, "arm-pose-estimation_amd", "lib", "diag", "libape_hip_testhooks.so")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", args[0]), *args[1:]], capture_output=True, text=True, timeout=300, env=env)
    line = [ln for ln in r.stdout.splitlines() if ln.strip()][-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and line and "OFF" not in line, (r.stdout[-600:], r.stderr[-600:])
    assert "vs oracle" in line, line
    kernel = {"6": "cluster32", "64": "cluster", "50": "upper128", "25": "upper32"}[args[3] if args[0] == "cold_stress.py" else args[3]]
    assert kernel in line, line
