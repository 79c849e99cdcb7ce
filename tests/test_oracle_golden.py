"""Pin the CPU oracle (oracle/ape_oracle.py) against golden vectors produced by the reference
itself (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import ape_oracle as orc

F64_TOL = 1e-12       # f64 numpy restatement vs f64 numpy reference
LSTM_TOL = 2e-6       # f32 cell loop vs torch.lstm (SURVEY 3.3 measured 1.1e-8 at T=6; T=64 drifts)


def quat_close(a, b, tol, sign_free_below=1e-4):
    """strict where |w_ref| is clear of zero; sign-aware only where the reference w ~ 0."""
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    d_plus = np.abs(a - b).max(axis=1)
    d_minus = np.abs(a + b).max(axis=1)
    amb = np.abs(b[:, 0]) < sign_free_below
    d = np.where(amb, np.minimum(d_plus, d_minus), d_plus)
    return np.nanmax(d) <= tol, float(np.nanmax(d))


# ---------------- primitives --------------------------------------------------------------
def test_quat_primitives(golden):
    g = golden("quat_ops.npz")
    assert np.allclose(orc.quat_mul(g["ham_a"], g["ham_b"]), g["ham"], rtol=0, atol=F64_TOL)
    assert np.allclose(orc.quat_rotate(g["rot_q"], g["rot_v"]), g["rot_out"], rtol=0, atol=F64_TOL)
    assert np.allclose(orc.quat_rotate(g["rot_q"], g["rot_single_v"]), g["rot_single_out"], rtol=0, atol=F64_TOL)
    hq = orc.hips_sin_cos_to_quat(g["hips_sin"], g["hips_cos"])
    assert np.allclose(hq, g["hips_quat"], rtol=0, atol=F64_TOL)
    # atan2(0, 0) = 0 -> identity; atan2(0, -1) = pi -> [cos(pi/2), 0, 1, 0]
    assert np.array_equal(hq[0], [1.0, 0.0, 0.0, 0.0])
    assert abs(hq[1, 2] - 1.0) < 1e-15


def test_six_drr_to_rotmat(golden):
    g = golden("quat_ops.npz")
    r = orc.six_drr_to_rotmat(g["six"])
    assert np.array_equal(np.isnan(r), np.isnan(g["rotmat"]))     # degenerate rows stay NaN
    assert np.allclose(r, g["rotmat"], rtol=0, atol=F64_TOL, equal_nan=True)


@pytest.mark.parametrize("route", ["eigh", "closed"])
def test_rotmat_to_quat(golden, route):
    g = golden("quat_ops.npz")
    ok_rows = ~np.isnan(g["quat"]).any(axis=1) & ~np.isnan(g["rotmat"]).any(axis=1)
    fn = orc.rotmat_to_quat_eigh if route == "eigh" else orc.rotmat_to_quat_closed
    q = fn(g["rotmat"][ok_rows])
    ok, worst = quat_close(q, g["quat"][ok_rows], 1e-9 if route == "closed" else F64_TOL)
    assert ok, worst
    assert (q[:, 0] >= 0).all()                                    # w >= 0 convention


def test_average_quaternions(golden):
    g = golden("quat_ops.npz")
    for i in range(4):
        out = orc.average_quaternions(g[f"avg_in_{i}"])
        assert np.allclose(out, g[f"avg_out_{i}"], rtol=0, atol=F64_TOL)


# ---------------- LSTM --------------------------------------------------------------------
@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_lstm_forward(golden, name):
    g = golden(f"lstm_{name}.npz")
    cfg = orc.MODEL_CONFIGS[name]
    for seed in (0, 1):
        sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], seed)
        assert np.array_equal(orc.state_dict_digest(sd), g[f"digest_seed{seed}"])
        for (B, T) in ((1, cfg["T"]), (5, cfg["T"]), (3, 64), (2, 1)):
            x, y_ref = g[f"x_seed{seed}_B{B}_T{T}"], g[f"y_seed{seed}_B{B}_T{T}"]
            y = orc.lstm_forward(sd, x)
            assert y.shape == y_ref.shape == (B, T, cfg["O"])
            assert np.abs(y - y_ref).max() < LSTM_TOL
            if (B, T) == (5, cfg["T"]):
                yt = orc.torch_reference_model(sd)(x)      # the third-party dependency itself
                assert np.abs(yt - y_ref).max() == 0.0


def test_ff_forward(golden):
    """MLP regressor (DropoutFF) restatement vs the reference module in eval mode"""
    g = golden("ff.npz")
    for tag in ("pocket_like", "small", "deep"):
        I, H, n_hidden, O = (int(v) for v in g["dims_" + tag])
        for seed in (0, 1):
            sd = orc.make_ff_state_dict(I, H, n_hidden, O, seed)
            for shape in ((1, 6, I), (37, 6, I), (300, I)):
                key = f"{tag}_seed{seed}_" + "x".join(map(str, shape))
                y = orc.ff_forward(sd, g["x_" + key])
                assert y.shape == g["y_" + key].shape
                assert np.abs(y - g["y_" + key]).max() < 2e-6


def test_imupose_forward(golden):
    """ImuPoseLSTM restatement (Linear+ReLU in front of a fixed 2 x 256 LSTM) vs the reference module"""
    g = golden("imupose.npz")
    for tag in ("pocket_like", "uarm_like"):
        I, O = (int(v) for v in g["dims_" + tag])
        for seed in (0, 1):
            sd = orc.make_imupose_state_dict(I, O, seed)
            assert np.array_equal(orc.state_dict_digest(sd), g[f"digest_{tag}_seed{seed}"])
            for (B, T) in ((1, 6), (21, 6), (3, 64), (2, 1)):
                key = f"{tag}_seed{seed}_B{B}_T{T}"
                y = orc.imupose_forward(sd, g["x_" + key])
                assert y.shape == g["y_" + key].shape == (B, T, O)
                assert np.abs(y - g["y_" + key]).max() < LSTM_TOL
                # its "Monte-Carlo" predictions are the plain forward of the one window, not n repeats
                assert g["ymc_" + key].shape == (1, T, O) and np.abs(g["ymc_" + key] - g["y_" + key][:1]).max() < 1e-6


def test_lstm_masks_are_interlayer_only():
    cfg = orc.MODEL_CONFIGS["uarm"]
    sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 5)
    x = np.random.default_rng(0).normal(size=(2, 4, cfg["I"])).astype(np.float32)
    ones = [np.ones((2, 4, cfg["H"]), np.float32)] * (cfg["L"] - 1)
    assert np.array_equal(orc.lstm_forward(sd, x, masks=ones), orc.lstm_forward(sd, x))
    zeros = [np.zeros((2, 4, cfg["H"]), np.float32)] * (cfg["L"] - 1)
    y0 = orc.lstm_forward(sd, x, masks=zeros)
    # with layer-0 output zeroed the result no longer depends on x
    assert np.array_equal(y0, orc.lstm_forward(sd, x * 0 + 1, masks=zeros))


# ---------------- FK + message --------------------------------------------------------------
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("route", ["eigh", "closed"])
def test_fk_and_msg(golden, layout, route):
    g = golden(f"fk_layout{layout}.npz")
    W = orc.LAYOUT_EST_WIDTH[layout]
    qcols = {0: (9, 13, 17), 1: (6, 10), 2: (9, 13, 17)}[layout]
    tol = F64_TOL if route == "eigh" else 1e-9
    for tag in ("bd", "bo"):
        body = g[f"body_{tag}"]
        for N in (1, 7, 300):
            preds, est_ref, msg_ref = g[f"preds_{tag}_N{N}"], g[f"est_{tag}_N{N}"], g[f"msg_{tag}_N{N}"]
            est = orc.arm_pose_from_targets(preds, body, layout, route)
            assert est.shape == (N, W)
            good = ~np.isnan(est_ref).any(axis=1)
            assert np.array_equal(good, ~np.isnan(est).any(axis=1))
            for c in qcols:
                ok, worst = quat_close(est[good, c:c + 4], est_ref[good, c:c + 4], tol)
                assert ok, (layout, tag, N, c, worst)
            # origins: strict wherever no quaternion of that row is sign-ambiguous
            clear = good & (np.abs(est_ref[:, [c for c in qcols]]) > 1e-4).all(axis=1)
            assert np.allclose(est[clear, :qcols[0]], est_ref[clear, :qcols[0]], rtol=0, atol=tol)
            # message from the REFERENCE est rows (isolates compose_msg bookkeeping)
            msg = orc.msg_from_est(est_ref[good] if N > 1 else est_ref, body, layout)
            if N == 1 or good.all():
                assert msg.shape == (25,)
                assert np.allclose(msg, msg_ref, rtol=0, atol=F64_TOL, equal_nan=True)
                assert np.array_equal(msg[0:4], msg[7:11])          # hand rot duplicates larm rot
                if layout == 1:
                    assert np.array_equal(msg[21:25], [1.0, 0.0, 0.0, 0.0])
                    assert np.array_equal(msg[18:21], body[0, 6:9])


# ---------------- streaming bookkeeping -------------------------------------------------------
@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_stream_trace(golden, norm_stats, name):
    g = golden(f"stream_trace_{name}.npz")
    cfg = orc.MODEL_CONFIGS[name]
    sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], int(g["weights_seed"]))
    stats = norm_stats[name]
    body = g["body"]
    assert int(g["seq_len"]) == cfg["T"]
    for smooth, mc in ((1, 1), (5, 1), (3, 4)):
        tag = f"s{smooth}_mc{mc}"

        def predict(hist):
            x = np.asarray(hist, dtype=np.float32)[None]
            y = orc.lstm_forward(sd, np.repeat(x, mc, axis=0))        # nn_models.py:206
            return y[:, -1, :]

        win = orc.WindowOracle(cfg["T"], smooth, stats, predict)
        for f, xx in enumerate(g[f"xx_{tag}"]):
            xx = xx.astype(np.float32) if str(g[f"xx_dtype_{tag}"]) == "float32" else xx
            pred = win.push(xx)
            pred_ref = g[f"pred_{tag}"][f]
            assert pred.shape == pred_ref.shape == (max(1, smooth if smooth > 1 else 1) * mc, cfg["O"])
            assert np.abs(pred - pred_ref).max() < 5e-6
            est = orc.arm_pose_from_targets(pred_ref, body, cfg["layout"], "eigh")
            msg = orc.msg_with_mc_samples(orc.msg_from_est(est, body, cfg["layout"]), est, True)
            msg_ref = g[f"msg_{tag}"][f]
            n_rows = pred_ref.shape[0]
            assert len(msg) == (25 + 6 * n_rows if n_rows > 1 else 25) == len(msg_ref)
            assert np.allclose(np.asarray(msg), msg_ref, rtol=0, atol=1e-11)


# ---------------- non-zero initial state and Monte-Carlo statistics vs the reference's own outputs ----------------
@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_lstm_initial_state_vs_reference_golden(golden, name):
    """DropoutLSTM.forward(x, hs=(h0, c0)) (nn_models.py:180-189): the cell loop started from the given state"""
    g = golden("lstm_hs.npz")
    cfg = orc.MODEL_CONFIGS[name]
    sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
    for (B, T) in ((1, cfg["T"]), (5, cfg["T"]), (3, 64), (18, 2)):
        k = f"{name}_B{B}_T{T}"
        y = orc.lstm_forward(sd, g["x_" + k], hs=(g["h0_" + k], g["c0_" + k]))
        assert np.abs(y - g["y_" + k]).max() < 2e-6
        assert np.abs(orc.lstm_forward(sd, g["x_" + k]) - g["y_" + k]).max() > 1e-3      # the state matters


def _oracle_mc_samples(sd, cfg, x, n, p, rng, scale=None, all_layers=False):
    """n Monte-Carlo samples of the last step through the oracle's masked cell loop: Bernoulli(1-p) masks scaled by
    1/(1-p) on the output sequence of every layer below the top one (what nn.LSTM(dropout=p) does in train mode)"""
    scale = 1.0 / (1.0 - p) if scale is None else scale
    T = x.shape[0]
    masks = [(rng.random((n, T, cfg["H"])) >= p).astype(np.float32) * np.float32(scale) for _ in range(cfg["L"] - 1)]
    return orc.lstm_forward(sd, np.repeat(x[None], n, axis=0), masks=masks)[:, -1, :]


@pytest.mark.parametrize("name", ["pocket", "uarm"])
def test_oracle_mc_dropout_matches_reference_distribution(golden, name):
    """the oracle's injected-mask loop with Bernoulli masks reproduces the DISTRIBUTION of the reference's
    `monte_carlo_predictions` samples (nn_models.py:191-207) -- and a wrong scale, rate or placement does not"""
    from tests import mc_check
    g = golden("mc_stats.npz")
    cfg = orc.MODEL_CONFIGS[name]
    sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], 0)
    p, n_ref, levels = float(g[f"dropout_{name}"]), int(g["n_samples"]), g["quantile_levels"]
    rng = np.random.default_rng(77)
    n = 8000
    for w in (0, 2):
        x = g[f"x_{name}"][w]
        args = (g[f"y_mean_{name}"][w], g[f"y_cov_{name}"][w], g[f"y_quant_{name}"][w], levels, n_ref)
        ok = mc_check.compare(_oracle_mc_samples(sd, cfg, x, n, p, rng), *args, what=f"{name} w{w}")
        assert not ok, ok
        if w == 0:       # negative controls: each must be flagged
            assert mc_check.compare(_oracle_mc_samples(sd, cfg, x, n, p, rng, scale=1.0), *args)          # no 1/(1-p)
            assert mc_check.compare(_oracle_mc_samples(sd, cfg, x, n, 0.5 * p, rng), *args)               # wrong rate
            assert mc_check.compare(orc.lstm_forward(sd, np.repeat(x[None], 64, axis=0))[:, -1, :] +
                                    np.zeros((64, 1), np.float32), *args)                                  # no dropout


@pytest.mark.parametrize("name", ["pocket", "uarm"])
def test_oracle_consumer_loop_mc_matches_reference_estimators(golden, norm_stats, name):
    """tests/golden/trace_mc_stats.npz (the reference ESTIMATORS in Monte-Carlo mode over the 20-row trace, hand / elbow rows
    of the last frame) against the oracle's chain for that frame: window of the last T feature rows, float64 z-score, masked
    cell loop with Bernoulli masks, de-normalisation, FK -- pins the fixture the GPU consumer-loop test is held to"""
    from tests import mc_check
    g, tr = golden("trace_mc_stats.npz"), golden(f"stream_trace_{name}.npz")
    cfg = orc.MODEL_CONFIGS[name]
    sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], int(g["weights_seed"]))
    st = norm_stats[name]
    xx = tr["xx_s1_mc1"][-cfg["T"]:]
    xx = xx.astype(np.float32) if str(tr["xx_dtype_s1_mc1"]) == "float32" else xx
    x = ((xx - st["xx_m"]) / st["xx_s"]).astype(np.float32)                          # estimator.py:103-104 + the float32 cast
    p, n_ref, levels = float(g[f"dropout_{name}"]), int(g["n_samples"]), g["quantile_levels"]
    rng = np.random.default_rng(78)
    y = _oracle_mc_samples(sd, cfg, x, 8000, p, rng).astype(np.float64) * st["yy_s"] + st["yy_m"]
    est6 = orc.arm_pose_from_targets(y, tr["body"], cfg["layout"], "closed")[:, :6]
    args = (g[f"est6_mean_{name}"], g[f"est6_cov_{name}"], g[f"est6_quant_{name}"], levels, n_ref)
    bad = mc_check.compare(est6, *args, what=f"{name} trace")
    assert not bad, bad
    y_bad = _oracle_mc_samples(sd, cfg, x, 8000, 0.5 * p, rng).astype(np.float64) * st["yy_s"] + st["yy_m"]
    assert mc_check.compare(orc.arm_pose_from_targets(y_bad, tr["body"], cfg["layout"], "closed")[:, :6], *args)
