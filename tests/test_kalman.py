"""Ensemble Kalman estimator (tail): HIP path (``ape_kalman_*`` through the host mirror of
``KalmanSmartwatchModel``) against ``oracle/kalman_oracle.py``.  PARITY UNPINNED -- the oracle restates the reference and the
published LinearFlipout algorithm but nothing of the reference could be run for this path (see the oracle's header); these
tests therefore prove HIP == restatement, not HIP == reference.

Tolerances (float32 path, stated per test): layer outputs and the corrected ensemble within 2e-4 absolute of the oracle on
injected draws (the oracle inverts the 14 x 14 innovation in float64, the kernel by float32 Gauss-Jordan with partial
pivoting); device-side draws (Philox) only statistically."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import kalman_oracle as ko
from tests import mc_check

TOL = 2e-4


def pack_noise(nz, order=ko.FLIPOUT_LAYERS):
    return np.concatenate([np.concatenate([nz[n]["eps_w"].ravel(), nz[n]["eps_b"].ravel(), nz[n]["sign_in"].ravel(),
                                           nz[n]["sign_out"].ravel()]) for n in order]).astype(np.float32)


def synthetic_inputs(rng, S, E, W):
    raw = rng.normal(size=(S, W, 1, 22)).astype(np.float32)
    state = (0.3 * rng.normal(size=(S, 1, W, 14)) + 0.05 * rng.normal(size=(S, E, W, 14))).astype(np.float32)
    return raw, state


# ---------------- oracle-only checks (CPU) ----------------------------------------------------------------------------------
def test_oracle_flipout_reduces_to_linear_without_perturbation():
    """eps = 0 -> LinearFlipout is the plain linear layer of its means; the perturbation is linear in eps and odd in the signs"""
    rng = np.random.default_rng(0)
    sd = ko.make_state_dict(10, 1)
    x = rng.normal(size=(32, 256)).astype(np.float32)
    nz = ko.draw_noise(rng, 10, 32)["sensor_model.fc3"]
    zero = dict(nz, eps_w=np.zeros_like(nz["eps_w"]), eps_b=np.zeros_like(nz["eps_b"]))
    base = ko.linear(x, sd["sensor_model.fc3.mu_weight"], sd["sensor_model.fc3.mu_bias"])
    assert np.array_equal(ko.linear_flipout(x, sd, "sensor_model.fc3", zero), base)
    full = ko.linear_flipout(x, sd, "sensor_model.fc3", nz)
    flipped = ko.linear_flipout(x, sd, "sensor_model.fc3", dict(nz, sign_out=-nz["sign_out"]))
    assert np.allclose(full - base, -(flipped - base), atol=1e-6)
    doubled = ko.linear_flipout(x, sd, "sensor_model.fc3", dict(nz, eps_w=2 * nz["eps_w"], eps_b=2 * nz["eps_b"]))
    assert np.allclose(doubled - base, 2 * (full - base), atol=1e-5)
    # perturbation scale: softplus(-3) ~ 0.0486 per weight -> std of a 256-term sum ~ 0.0486 * |x| ~ 0.78 for unit inputs
    assert 0.3 < float(np.std(full - base)) < 1.5


def test_oracle_update_limits():
    """the Kalman update: a huge observation noise leaves the prediction alone, a tiny one moves every member's state onto its
    observation inside the ensemble's subspace (gain -> P P^+); the corrected mean lies between prediction and observation"""
    rng = np.random.default_rng(1)
    sd = ko.make_state_dict(10, 2)
    raw, state = synthetic_inputs(rng, 1, 32, 10)
    nz = ko.draw_noise(rng, 10, 32)
    out = ko.kalman_forward(sd, raw, state, nz)
    assert [o.shape for o in out] == [(1, 32, 14), (1, 1, 14), (1, 1, 14), (1, 1, 14), (1, 32, 14)]
    lo, hi = np.minimum(out[2], out[3]) - 0.5, np.maximum(out[2], out[3]) + 0.5
    assert np.all((out[1] > lo) & (out[1] < hi))
    big = dict(sd)
    big["observation_noise.fc2.bias"] = sd["observation_noise.fc2.bias"] + np.float32(1e4)
    out_big = ko.kalman_forward(big, raw, state, nz)
    pred = ko.process_model(sd, state, nz)
    assert np.abs(out_big[0] - pred).max() < 1e-3


def test_oracle_frame_logic():
    """watch_phone_pocket_kalman.py:133-169: win_size + 1 sensor-only frames ([1,14]), then the ensemble ([E,14])"""
    rng = np.random.default_rng(2)
    E, W = 32, 10
    fl = ko.KalmanFrameLogic(ko.make_state_dict(W, 3), E, W)
    shapes = []
    for f in range(W + 4):
        y = fl.step(rng.normal(size=(W, 22)), ko.draw_noise(rng, W, E), rng.standard_normal((E, 14)))
        shapes.append(y.shape)
        assert np.all(np.isfinite(y))
    assert shapes == [(1, 14)] * (W + 1) + [(E, 14)] * 3
    assert fl.state.shape == (1, E, W, 14) and np.abs(fl.state[0, :, 0]).max() > 0       # the zero history has been pushed out


# ---------------- HIP vs oracle (GPU) -----------------------------------------------------------------------------------------
def make_model(E, W, seed=0):
    from wear_mocap_ape_amd.estimate import kalman_models
    sd = ko.make_state_dict(W, seed)
    m = kalman_models.KalmanSmartwatchModel(E, W)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m, sd


@pytest.mark.gpu
@pytest.mark.parametrize("S,E,W", [(1, 32, 10), (1, 48, 10), (3, 32, 10), (2, 16, 4), (1, 40, 6)])
def test_forward_injected_draws(S, E, W):
    """every output of KalmanSmartwatchModel.forward on injected draws, within 2e-4 of the oracle"""
    rng = np.random.default_rng(10 * S + E)
    m, sd = make_model(E, W, 4)
    for rep in range(2):
        raw, state = synthetic_inputs(rng, S, E, W)
        nz = ko.draw_noise(rng, W, S * E)
        blob = pack_noise(nz)
        assert blob.size == m.noise_floats(S)
        got = [t.cpu().numpy() for t in m.forward(torch.from_numpy(raw), torch.from_numpy(state), noise=torch.from_numpy(blob))]
        want = ko.kalman_forward(sd, raw, state, nz)
        for name, g, w in zip(("state_corrected", "m_state_corrected", "m_state_pred", "z", "ensemble_z"), got, want):
            assert g.shape == w.shape, name
            assert np.abs(g - w).max() < TOL, (name, S, E, W, float(np.abs(g - w).max()))
    m.check()


@pytest.mark.gpu
def test_forward_is_deterministic_per_seed_and_rejects_bad_shapes():
    m, _ = make_model(32, 10, 5)
    rng = np.random.default_rng(0)
    raw, state = synthetic_inputs(rng, 1, 32, 10)
    a = [t.cpu().numpy() for t in m.manual_seed(7).forward(raw, state)]
    b = [t.cpu().numpy() for t in m.manual_seed(7).forward(raw, state)]
    c = [t.cpu().numpy() for t in m.manual_seed(8).forward(raw, state)]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert not np.array_equal(a[0], c[0])
    with pytest.raises(UserWarning):
        m.forward(raw[:, :5], state)
    with pytest.raises(UserWarning):
        m.forward(raw, state[:, :16])
    from wear_mocap_ape_amd.estimate import kalman_models
    with pytest.raises(UserWarning):
        kalman_models.KalmanSmartwatchModel(32, 7)            # odd window
    with pytest.raises(UserWarning):
        kalman_models.KalmanSmartwatchModel(32, 10).forward(raw, state)      # weights not loaded
    sd = ko.make_state_dict(10, 0)
    del sd["sensor_model.fc5.rho_bias"]
    with pytest.raises(UserWarning):
        kalman_models.KalmanSmartwatchModel(32, 10).load_state_dict(sd)


@pytest.mark.gpu
def test_format_state():
    """kalman_models.py:164-173: state + sqrt(0.1) * N(0, I); injected draws exactly, device draws statistically"""
    m, _ = make_model(48, 10, 6)
    rng = np.random.default_rng(3)
    st = rng.normal(size=(1, 14)).astype(np.float32)
    nzs = rng.standard_normal((48, 14)).astype(np.float32)
    got = m.format_state(torch.from_numpy(st), noise=torch.from_numpy(nzs)).cpu().numpy()
    assert np.abs(got - ko.format_state(st, nzs)).max() < 1e-6
    draws = np.concatenate([m.format_state(torch.from_numpy(st)).cpu().numpy() - st for _ in range(200)])     # [9600, 14]
    n = draws.shape[0]
    assert np.abs(draws.mean(axis=0)).max() < 5.5 * np.sqrt(0.1 / n)
    assert np.abs(draws.var(axis=0) / 0.1 - 1).max() < 5.5 * np.sqrt(2.0 / n)
    c = np.corrcoef(draws, rowvar=False) - np.eye(14)
    assert np.abs(c).max() < 5.5 / np.sqrt(n)
    # Box-Muller tails: |x| > 3 sigma in 0.27 % of the draws
    frac = float(np.mean(np.abs(draws) > 3 * np.sqrt(0.1)))
    assert abs(frac - 0.0026998) < 5.5 * np.sqrt(0.0027 / draws.size)


@pytest.mark.gpu
def test_device_draws_match_the_oracles_distribution():
    """forward with device-side (Philox) draws: the distribution of ensemble_z and state_corrected rows over many calls against
    the oracle's under numpy draws (mean / std / quantile fractions at 5.5 standard errors, tests/mc_check.py)"""
    E, W = 32, 10
    m, sd = make_model(E, W, 7)
    rng = np.random.default_rng(11)
    raw, state = synthetic_inputs(rng, 1, E, W)
    n_ref_calls, n_calls = 260, 260
    ref = [ko.kalman_forward(sd, raw, state, ko.draw_noise(rng, W, E)) for _ in range(n_ref_calls)]
    m.manual_seed(1234)
    rt, stt = torch.from_numpy(raw).cuda(), torch.from_numpy(state).cuda()
    got = [[t.cpu().numpy() for t in m.forward(rt, stt)] for _ in range(n_calls)]
    levels = np.array([0.1, 0.5, 0.9])
    for idx, name in ((4, "ensemble_z"), (0, "state_corrected")):
        r = np.concatenate([o[idx][0] for o in ref]).astype(np.float64)          # [n_ref_calls * E, 14]
        g = np.concatenate([o[idx][0] for o in got]).astype(np.float64)
        # rows of one call share the weight perturbation: the effective sample size for the mean is the number of calls, not
        # of rows -- compare per-call means for location, pooled rows for spread
        bad = mc_check.compare(np.stack([o[idx][0].mean(axis=0) for o in got]), np.stack([o[idx][0].mean(axis=0) for o in ref]).mean(axis=0),
                               np.cov(np.stack([o[idx][0].mean(axis=0) for o in ref]), rowvar=False),
                               np.quantile(np.stack([o[idx][0].mean(axis=0) for o in ref]), levels, axis=0), levels, n_ref_calls, name + " call means")
        assert not bad, bad
        ratio = g.std(axis=0) / r.std(axis=0)
        assert np.abs(ratio - 1).max() < 0.12, (name, ratio)
    # negative control: a sampler without the sign flips (rows of a call perfectly correlated) has no within-call spread
    within_ref = float(np.mean([o[4][0].std(axis=0).mean() for o in ref]))
    within_got = float(np.mean([o[4][0].std(axis=0).mean() for o in got]))
    assert abs(within_got / within_ref - 1) < 0.1 and within_ref > 0.05
    m.check()


@pytest.mark.gpu
def test_estimator_frame_logic_vs_oracle():
    """WatchPhonePocketKalman.make_prediction_from_row_hist over 14 frames with the draws of every call injected: the same
    [1,14] / [E,14] outputs as the oracle's frame logic; then the public loop: message lengths 25 and 25 + 6 E"""
    from wear_mocap_ape_amd.estimate.watch_phone_pocket_kalman import WatchPhonePocketKalman
    E, W = 32, 10
    sd = ko.make_state_dict(W, 8)
    est = WatchPhonePocketKalman({k: torch.from_numpy(v) for k, v in sd.items()}, smooth=1, num_ensemble=E, window_size=W,
                                 normalize=False)
    fl = ko.KalmanFrameLogic(sd, E, W)
    rng = np.random.default_rng(12)
    model = est.model
    fwd, fmt = model.forward, model.format_state
    cur = {}
    model.forward = lambda ro, sp: fwd(ro, sp, noise=torch.from_numpy(pack_noise(cur["nz"])))
    model.__class__.__call__ = lambda self, ro, sp: self.forward(ro, sp)
    model.format_state = lambda st: fmt(st, noise=torch.from_numpy(cur["init"]))
    try:
        for f in range(W + 4):
            cur["nz"], cur["init"] = ko.draw_noise(rng, W, E), rng.standard_normal((E, 14)).astype(np.float32)
            hist = rng.normal(size=(W, 22))
            got = est.make_prediction_from_row_hist(hist)
            want = fl.step(hist, cur["nz"], cur["init"])
            assert got.shape == want.shape == ((1, 14) if f <= W else (E, 14))
            assert np.abs(got - want).max() < 5e-4, (f, float(np.abs(got - want).max()))      # 14 frames of feedback
    finally:
        model.__class__.__call__ = model.__class__.forward
    # the public path: rows in, messages out (estimator.py:131-137: 25 floats + 6 per stacked row when there are several)
    est.reset()
    msgs = []
    row = np.zeros(55, dtype=np.float32)
    row[[5, 28, 46, 50]] = 1.0                    # unit quaternions (w) for the rotation vectors and the forward calibrations
    for f in range(W + 3):
        pred = est.add_xx_to_row_hist_and_make_prediction(est.parse_row_to_xx(row + 0.01 * f))
        msgs.append(est.msg_from_pred(pred, True))
    assert [len(mm) for mm in msgs] == [25] * (W + 1) + [25 + 6 * E] * 2
    assert all(np.all(np.isfinite(np.asarray(mm, dtype=np.float64))) for mm in msgs)
    est.model.check()
