"""CPU oracle for the ensemble Kalman estimator  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A numpy restatement of ``KalmanSmartwatchModel`` (reference ``estimate/kalman_models.py``, cited as
``kalman_models.py:<line>`` relative to ``/root/reference/src/wear_mocap_ape/estimate``) and of the frame logic of
``WatchPhonePocketKalman.make_prediction_from_row_hist`` (``watch_phone_pocket_kalman.py:133-169``): SURVEY.md section 8
row f4 (tail).  Only ``tests/`` may import this module, and only as the checker.

Parity status: PARITY UNPINNED.  The reference module cannot be imported here: its first line imports
``bayesian_torch.layers.flipout_layers.linear_flipout.LinearFlipout`` (kalman_models.py:1; setup.cfg:30 lists
``bayesian_torch`` without a version) and that package is not installed; the trained checkpoint
(``data_deploy/kalman/SW-v3.8-model-436400``) is absent (``.MISSING_LARGE_BLOBS``), and the reference holds no test,
fixture or golden vector for this path.  No stand-in for the missing library was written to make the reference run.
What is restated:

  * ``LinearFlipout.forward`` of bayesian-torch (Intel Labs; published algorithm, Wen et al. 2018 "Flipout"), as the
    reference calls it (``x, _ = layer(x)``, kalman_models.py:44,46,124,126,128):
        sigma_W = log1p(exp(rho_W));  delta_W = sigma_W * eps_W,  eps_W ~ N(0,1) drawn once per call (shared by all rows)
        bias perturbation likewise from (rho_b, eps_b)
        out = x @ mu_W^T + mu_b
        sign_in ~ sign(U(-1,1)) of x's shape, sign_out ~ sign(U(-1,1)) of out's shape  (one per element: per row)
        result = out + ((x * sign_in) @ delta_W^T + delta_b) * sign_out
    parameter names ``mu_weight, rho_weight, mu_bias, rho_bias`` (its state_dict keys; the ``eps_*`` / ``prior_*`` buffers
    of a checkpoint are ignored: eps is redrawn on every call).
  * everything else from the reference's own lines, cited at each function.

Every random draw is an explicit argument (``noise``), so that the HIP path can be compared with injected draws bit for
tolerance; ``draw_noise`` makes them from a numpy generator for the statistical tests.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

F = np.float32
DIM_X = 14                 # kalman_models.py:146 (dim_x = dim_z = 14)
RAW_OBS = 22               # kalman_models.py:146
FLIPOUT_LAYERS = ("process_model.bayes1", "process_model.bayes3", "sensor_model.fc3", "sensor_model.fc5", "sensor_model.fc6")


def layer_shapes(win_size: int) -> Dict[str, tuple]:
    """(out_features, in_features) of every layer, reference names (kalman_models.py:32-34, 70-71, 112-115)"""
    return {
        "process_model.bayes1": (256, DIM_X * win_size), "process_model.bayes3": (512, 256), "process_model.bayes_m2": (DIM_X, 512),
        "sensor_model.fc2": (256, RAW_OBS * win_size), "sensor_model.fc3": (256, 256), "sensor_model.fc5": (64, 256),
        "sensor_model.fc6": (DIM_X, 64),
        "observation_noise.fc1": (32, DIM_X), "observation_noise.fc2": (DIM_X, 32),
    }


def make_state_dict(win_size: int = 10, seed: int = 0) -> Dict[str, np.ndarray]:
    """seeded synthetic parameters in the reference's state_dict layout (``checkpoint["model"]``,
    watch_phone_pocket_kalman.py:50-54): mu ~ U(+-1/sqrt(fan_in)) like torch's Linear, rho ~ N(-3, 0.1) (bayesian-torch's
    default posterior_rho_init = -3)"""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, (n, k) in layer_shapes(win_size).items():
        b = 1.0 / np.sqrt(k)
        if name in FLIPOUT_LAYERS:
            sd[name + ".mu_weight"] = rng.uniform(-b, b, (n, k)).astype(F)
            sd[name + ".rho_weight"] = (-3.0 + 0.1 * rng.standard_normal((n, k))).astype(F)
            sd[name + ".mu_bias"] = rng.uniform(-b, b, n).astype(F)
            sd[name + ".rho_bias"] = (-3.0 + 0.1 * rng.standard_normal(n)).astype(F)
        else:
            sd[name + ".weight"] = rng.uniform(-b, b, (n, k)).astype(F)
            sd[name + ".bias"] = rng.uniform(-b, b, n).astype(F)
    return sd


def softplus(x):
    return np.log1p(np.exp(x.astype(F))).astype(F)


def leaky_relu(x, slope=0.01):          # torch.nn.functional.leaky_relu default negative_slope
    return np.where(x >= 0, x, F(slope) * x).astype(F)


def linear(x, w, b):
    return (x.astype(F) @ w.T.astype(F) + b.astype(F)).astype(F)


def linear_flipout(x, sd, name, nz):
    """bayesian-torch LinearFlipout.forward (see the module docstring); nz = {"eps_w" [N,K], "eps_b" [N], "sign_in" [R,K],
    "sign_out" [R,N]}"""
    out = linear(x, sd[name + ".mu_weight"], sd[name + ".mu_bias"])
    dw = (softplus(sd[name + ".rho_weight"]) * nz["eps_w"].astype(F)).astype(F)
    db = (softplus(sd[name + ".rho_bias"]) * nz["eps_b"].astype(F)).astype(F)
    pert = linear((x * nz["sign_in"]).astype(F), dw, db)
    return (out + pert * nz["sign_out"]).astype(F)


def draw_noise(rng, win_size: int, rows: int) -> Dict[str, Dict[str, np.ndarray]]:
    """one forward call's random draws; rows = batch * ensemble"""
    nz = {}
    for name in FLIPOUT_LAYERS:
        n, k = layer_shapes(win_size)[name]
        nz[name] = {"eps_w": rng.standard_normal((n, k)).astype(F), "eps_b": rng.standard_normal(n).astype(F),
                    "sign_in": np.where(rng.random((rows, k)) < 0.5, F(-1), F(1)).astype(F),
                    "sign_out": np.where(rng.random((rows, n)) < 0.5, F(-1), F(1)).astype(F)}
    return nz


def process_model(sd, x, nz):
    """kalman_models.py:37-50: x [bs, E, W, 14] -> [bs, E, 14]"""
    bs, E = x.shape[0], x.shape[1]
    h = x.reshape(bs * E, -1).astype(F)
    h = leaky_relu(linear_flipout(h, sd, "process_model.bayes1", nz["process_model.bayes1"]))
    h = leaky_relu(linear_flipout(h, sd, "process_model.bayes3", nz["process_model.bayes3"]))
    h = linear(h, sd["process_model.bayes_m2.weight"], sd["process_model.bayes_m2.bias"])
    return h.reshape(bs, E, DIM_X)


def sensor_model(sd, x, E, nz):
    """kalman_models.py:117-136: x [bs, W, 1, 22] -> (ensemble [bs, E, 14], mean [bs, 1, 14]).  NB the reference's
    ``x.repeat(E,1,1,1)`` + reshape to (bs*E, ...) orders the rows member-major (row = e*bs + b) while its final reshape
    reads them batch-major; for the deployed batch size 1 the two coincide.  Restated batch-major (row = b*E + e reads stream
    b), which is the reference for bs = 1 and the only consistent reading for a bank of streams."""
    bs = x.shape[0]
    h = np.repeat(x.reshape(bs, -1), E, axis=0).astype(F)
    h = leaky_relu(linear(h, sd["sensor_model.fc2.weight"], sd["sensor_model.fc2.bias"]))
    h = leaky_relu(linear_flipout(h, sd, "sensor_model.fc3", nz["sensor_model.fc3"]))
    h = leaky_relu(linear_flipout(h, sd, "sensor_model.fc5", nz["sensor_model.fc5"]))
    h = linear_flipout(h, sd, "sensor_model.fc6", nz["sensor_model.fc6"])
    ens = h.reshape(bs, E, DIM_X)
    return ens, ens.mean(axis=1, dtype=F)[:, None, :].astype(F)


def observation_noise(sd, z):
    """kalman_models.py:73-80: z [bs, 1, 14] -> diagonal of R [bs, 14] (the reference returns diag_embed of it)"""
    h = np.maximum(linear(z.reshape(-1, DIM_X), sd["observation_noise.fc1.weight"], sd["observation_noise.fc1.bias"]), 0).astype(F)
    h = linear(h, sd["observation_noise.fc2.weight"], sd["observation_noise.fc2.bias"])
    return (np.square(h + F(1e-3)) + F(0.038729833)).astype(F)


def kalman_forward(sd, raw_obs, state_prev, nz):
    """kalman_models.py:175-220 for batch size 1 per stream (``A = state_pred - state_m`` broadcasts [bs,E,14] - [bs,14],
    which only is the ensemble anomaly for bs = 1 -- the deployed case, watch_phone_pocket_kalman.py:135; a bank of
    streams is therefore restated as independent bs = 1 updates over a shared flipout draw).
    raw_obs [S, W, 1, 22], state_prev [S, E, W, 14] -> (state_corrected [S,E,14], m_state_corrected [S,1,14],
    m_state_pred [S,1,14], z [S,1,14], ensemble_z [S,E,14])"""
    S, E = state_prev.shape[0], state_prev.shape[1]
    state_pred = process_model(sd, state_prev, nz)                                     # :181
    state_m = state_pred.mean(axis=1, dtype=F)                                         # :183
    ens_z, z = sensor_model(sd, raw_obs, E, nz)                                        # :194
    r_diag = observation_noise(sd, z)                                                  # :198
    corrected = np.empty_like(state_pred)
    for s in range(S):
        A = (state_pred[s] - state_m[s]).astype(F)                                     # [E,14]  :184
        P = (F(1.0 / (E - 1)) * (A.T @ A)).astype(F)                                   # :200, :203
        innovation = (P + np.diag(r_diag[s])).astype(F)
        inv = np.linalg.inv(innovation.astype(np.float64)).astype(F)                   # :201 (float64 here: the checker)
        K = (P @ inv).astype(F)                                                        # :202-204
        gain = (K @ (ens_z[s].T - state_pred[s].T)).T.astype(F)                        # :206
        corrected[s] = state_pred[s] + gain                                            # :208
    return (corrected, corrected.mean(axis=1, dtype=F)[:, None, :], state_m[:, None, :], z, ens_z)


def format_state(state, init_noise):
    """kalman_models.py:164-173: state [k,14] (k = 1) -> repeat over the ensemble + N(0, 0.1 I) draw; init_noise [E,14] is the
    standard-normal draw (MultivariateNormal(0, 0.1 I).sample == sqrt(0.1) * N(0, I))"""
    E = init_noise.shape[0]
    return (np.tile(state, (E, 1)).astype(F) + (F(np.sqrt(F(0.1))) * init_noise.astype(F))).astype(F)


class KalmanFrameLogic:
    """watch_phone_pocket_kalman.py:57-63, 133-169 for one stream: the filter starts from a zero history; while
    ``init_step <= win_size`` a frame returns the sensor model's mean observation [1,14] and appends ``format_state`` of it;
    afterwards it returns the corrected ensemble [E,14] and appends it."""

    def __init__(self, sd, num_ensemble=32, win_size=10):
        self.sd, self.E, self.W = sd, num_ensemble, win_size
        self.state = np.zeros((1, num_ensemble, win_size, DIM_X), F)
        self.init_step = 0

    def step(self, xx_hist, nz, init_noise: Optional[np.ndarray] = None):
        raw = np.asarray(xx_hist, F)[None, :, None, :]                                # :135
        out = kalman_forward(self.sd, raw, self.state, nz)
        if self.init_step <= self.W:                                                    # :141
            self.init_step += 1
            pred = format_state(out[3][0], init_noise)[None, :, None, :]                # :145-146
            self.state = np.concatenate((self.state[:, :, 1:, :], pred), axis=2)
            return out[3][0][:, :14]                                                    # :152-156
        self.state = np.concatenate((self.state[:, :, 1:, :], out[0][:, :, None, :]), axis=2)   # :160-162
        return out[0][0][:, :14]                                                        # :169
