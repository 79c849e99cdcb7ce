#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run in the build container only (the reference does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden.py

What it does
  * imports ``wear_mocap_ape`` from /root/reference/src (read-only);
  * ``aenum`` (setup.cfg:23) is not installed here, so ``utility/names.py`` cannot import; an
    in-process stand-in for ``aenum.Enum`` / ``aenum.NoAlias`` (recipe of SURVEY.md appendix A:
    an ``enum.Enum`` whose class dict ignores ``_settings_``) is registered first.  It carries no
    arithmetic; transformations.py and nn_models.py import without it;
  * the trained checkpoints are absent (.MISSING_LARGE_BLOBS), so models are built from
    results.json and loaded with the seeded synthetic weights of ``oracle.make_state_dict``
    (numpy PCG64, platform-stable); tests regenerate the same weights from the seed and
    compare ``state_dict_digest``;
  * writes inputs + the reference's outputs as .npz/.json fixtures (data only).

Also exports the three normalisation-stat pickles and the fields of the three results.json
that the loader uses into the product package's ``data_deploy`` (model/config data).
"""
import enum
import json
from array import array
import os
import pickle
import sys
import types
import warnings
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
REF_SRC = Path("/root/reference/src")
OUT = REPO / "tests" / "golden"
PKG_DEPLOY = REPO / "arm-pose-estimation_amd" / "wear_mocap_ape_amd" / "data_deploy"

sys.dont_write_bytecode = True
sys.path.insert(0, str(REF_SRC))
sys.path.insert(0, str(REPO))


# ---- stand-in for the missing `aenum` dependency (no arithmetic) ------------------------
class _IgnoreSettingsDict(enum._EnumDict):
    def __setitem__(self, key, value):
        if key == "_settings_":
            return
        super().__setitem__(key, value)


class _Meta(enum.EnumMeta):
    @classmethod
    def __prepare__(mcs, cls, bases, **kw):
        base = super().__prepare__(cls, bases, **kw)
        d = _IgnoreSettingsDict()
        d.__dict__.update(base.__dict__)
        for k, v in base.items():
            dict.__setitem__(d, k, v)
        return d


class _Enum(enum.Enum, metaclass=_Meta):
    pass


_m = types.ModuleType("aenum")
_m.Enum = _Enum
_m.NoAlias = object()
sys.modules["aenum"] = _m

import torch  # noqa: E402

from oracle import ape_oracle as orc  # noqa: E402
import wear_mocap_ape.config as ref_config  # noqa: E402
import wear_mocap_ape.estimate.nn_models as ref_nn  # noqa: E402
import wear_mocap_ape.utility.transformations as ref_ts  # noqa: E402
from wear_mocap_ape.data_deploy.nn import deploy_models as ref_deploy  # noqa: E402
from wear_mocap_ape.estimate import compose_msg as ref_msg  # noqa: E402
from wear_mocap_ape.estimate import estimate_joints as ref_fk  # noqa: E402
from wear_mocap_ape.utility import data_stats as ref_stats  # noqa: E402
from wear_mocap_ape.utility.names import NNS_INPUTS, NNS_TARGETS  # noqa: E402

HASHES = {
    "pocket": ref_deploy.LSTM.WATCH_PHONE_POCKET.value,
    "watch": ref_deploy.LSTM.WATCH_ONLY.value,
    "uarm": ref_deploy.LSTM.WATCH_PHONE_UARM.value,
}
LAYOUT_OF = {
    "ORI_CAL_LARM_UARM_HIPS": orc.LAYOUT_ORI_CAL_LARM_UARM_HIPS,
    "ORI_CAL_LARM_UARM": orc.LAYOUT_ORI_CAL_LARM_UARM,
    "ORI_POS_CAL_LARM_UARM_HIPS": orc.LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS,
}


def ref_params(name):
    p = Path(ref_config.PATHS["deploy"]) / "nn" / HASHES[name] / "results.json"
    return json.loads(p.read_text())


def ref_model(name, seed, dropout=None):
    """reference DropoutLSTM (nn_models.py:160) with the oracle's seeded weights."""
    p = ref_params(name)
    assert p["model"] == "DropoutLSTM"
    model = ref_nn.DropoutLSTM(input_size=len(p["x_inputs_v"]), hidden_layer_size=p["hidden_layer_size"],
                               hidden_layer_count=p["hidden_layer_count"], output_size=len(p["y_targets_v"]),
                               dropout=p["dropout"] if dropout is None else dropout)
    sd = orc.make_state_dict(len(p["x_inputs_v"]), p["hidden_layer_size"], p["hidden_layer_count"],
                             len(p["y_targets_v"]), seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return model.eval(), p, sd


def rand_unit_quats(rng, n):
    q = rng.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


# ---- 1. norm stats + model config export ------------------------------------------------
def export_stats_and_configs():
    stats_out = {}
    for name in HASHES:
        p = ref_params(name)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            st = ref_stats.get_norm_stats(NNS_INPUTS[p["x_inputs_n"]], NNS_TARGETS[p["y_targets_n"]])
        fname = f"{p['x_inputs_n']}_{p['y_targets_n']}.json"
        plain = {k: np.asarray(st[k], dtype=np.float64).tolist() for k in ("xx_m", "xx_s", "yy_m", "yy_s")}
        plain["x_inputs"] = list(NNS_INPUTS[p["x_inputs_n"]].value)
        plain["y_targets"] = list(NNS_TARGETS[p["y_targets_n"]].value)
        (PKG_DEPLOY / "data_stats").mkdir(parents=True, exist_ok=True)
        (PKG_DEPLOY / "data_stats" / fname).write_text(json.dumps(plain, indent=1))
        stats_out[name] = plain
        keep = ("model", "hidden_layer_count", "hidden_layer_size", "dropout", "sequence_len", "normalize",
                "hash", "y_targets_n", "x_inputs_n", "y_targets_v", "x_inputs_v")
        d = PKG_DEPLOY / "nn" / HASHES[name]
        d.mkdir(parents=True, exist_ok=True)
        (d / "results.json").write_text(json.dumps({k: p[k] for k in keep}, indent=1))
    (OUT / "norm_stats.json").write_text(json.dumps(stats_out, indent=1))
    return stats_out


# ---- 2. LSTM forward goldens ------------------------------------------------------------
def gen_lstm(stats):
    for name, cfg in orc.MODEL_CONFIGS.items():
        blob = {}
        for seed in (0, 1):
            model, p, sd = ref_model(name, seed)
            blob[f"digest_seed{seed}"] = orc.state_dict_digest(sd)
            rng = np.random.default_rng(100 + seed)
            for (B, T) in ((1, cfg["T"]), (5, cfg["T"]), (3, 64), (2, 1)):
                x = rng.normal(size=(B, T, cfg["I"])).astype(np.float32)
                with torch.no_grad():
                    y = model(torch.from_numpy(x)).numpy()
                blob[f"x_seed{seed}_B{B}_T{T}"] = x
                blob[f"y_seed{seed}_B{B}_T{T}"] = y
        np.savez_compressed(OUT / f"lstm_{name}.npz", **blob)


# ---- 2b. non-zero initial state: DropoutLSTM.forward(x, hs=(h0, c0)) (nn_models.py:180-189) ----------
def gen_lstm_hs():
    blob = {}
    for name, cfg in orc.MODEL_CONFIGS.items():
        model, p, sd = ref_model(name, 0)
        rng = np.random.default_rng(400)
        for (B, T) in ((1, cfg["T"]), (5, cfg["T"]), (3, 64), (18, 2)):
            x = rng.normal(size=(B, T, cfg["I"])).astype(np.float32)
            h0 = (0.5 * rng.normal(size=(cfg["L"], B, cfg["H"]))).astype(np.float32)
            c0 = (0.7 * rng.normal(size=(cfg["L"], B, cfg["H"]))).astype(np.float32)
            with torch.no_grad():
                y = model(torch.from_numpy(x), (torch.from_numpy(h0), torch.from_numpy(c0))).numpy()
            for k, v in (("x", x), ("h0", h0), ("c0", c0), ("y", y)):
                blob[f"{k}_{name}_B{B}_T{T}"] = v
    np.savez_compressed(OUT / "lstm_hs.npz", **blob)


# ---- 2c. Monte-Carlo dropout statistics of the reference itself (nn_models.py:191-207) ---------------------
MC_QUANTILES = (0.05, 0.25, 0.5, 0.75, 0.95)
MC_SAMPLES = 24000


def gen_mc_stats():
    """`ref_model.monte_carlo_predictions(n, x)[:, -1]` -- self.lstm.train() + x.repeat, torch's own mask stream --
    drawn MC_SAMPLES times for three fixed windows per deployed model; mean, covariance and quantiles of the NN targets
    and of the FK'd hand / elbow positions (est[:, :6]).  The masks themselves cannot be replayed, their
    DISTRIBUTION can: scale 1/(1-p), placement between the layers only, every step, probability p."""
    blob = {"quantile_levels": np.array(MC_QUANTILES), "n_samples": np.array(MC_SAMPLES)}
    for name, cfg in orc.MODEL_CONFIGS.items():
        model, p, sd = ref_model(name, 0)
        tgt = NNS_TARGETS[p["y_targets_n"]]
        rng = np.random.default_rng(500)
        xs = rng.normal(size=(3, cfg["T"], cfg["I"])).astype(np.float32)
        xs[2] *= 2.5                                      # one window far from the mean (saturating gates)
        torch.manual_seed(1234)
        stats_y, stats_e = [], []
        for w in range(3):
            chunks = []
            with torch.no_grad():
                for _ in range(MC_SAMPLES // 4000):
                    chunks.append(model.monte_carlo_predictions(4000, torch.from_numpy(xs[w:w + 1])).numpy()[:, -1, :])
            y = np.concatenate(chunks).astype(np.float64)                # [n, O]
            est6 = ref_fk.arm_pose_from_nn_targets(y, orc.DEFAULT_BODY, tgt)[:, :6]
            for arr, dst in ((y, stats_y), (est6, stats_e)):
                dst.append((arr.mean(axis=0), np.cov(arr, rowvar=False), np.quantile(arr, MC_QUANTILES, axis=0)))
        assert model.lstm.training                      # nn_models.py:204: permanent
        blob[f"x_{name}"] = xs
        for tag, st in (("y", stats_y), ("est6", stats_e)):
            blob[f"{tag}_mean_{name}"] = np.array([s_[0] for s_ in st])
            blob[f"{tag}_cov_{name}"] = np.array([s_[1] for s_ in st])
            blob[f"{tag}_quant_{name}"] = np.array([s_[2] for s_ in st])
        blob[f"dropout_{name}"] = np.array(p["dropout"])
    # DropoutFF: dropout in front of the output layer (nn_models.py:356-370)
    I, H, n_hidden, O = 22, 256, 2, 14
    sdf = orc.make_ff_state_dict(I, H, n_hidden, O, 0)
    ff = ref_nn.DropoutFF(output_size=O, hidden_layer_size=H, hidden_layer_count=n_hidden, input_size=I, dropout=0.2)
    ff.load_state_dict({k: torch.from_numpy(v) for k, v in sdf.items()})
    ff.eval()
    xf = np.random.default_rng(501).normal(size=(1, 1, I)).astype(np.float32)
    torch.manual_seed(4321)
    with torch.no_grad():
        yf = ff.monte_carlo_predictions(MC_SAMPLES, torch.from_numpy(xf)).numpy()[:, -1, :].astype(np.float64)
    blob["x_ff"], blob["dims_ff"] = xf, np.array([I, H, n_hidden, O])
    blob["y_mean_ff"], blob["y_cov_ff"] = yf.mean(axis=0)[None], np.cov(yf, rowvar=False)[None]
    blob["y_quant_ff"] = np.quantile(yf, MC_QUANTILES, axis=0)[None]
    np.savez_compressed(OUT / "mc_stats.npz", **blob)


def gen_ff():
    """DropoutFF (the MLP regressor the loader can dispatch, nn_models.py:313-370,395-396) in eval mode"""
    blob = {}
    for tag, (I, H, n_hidden, O) in {"pocket_like": (22, 256, 2, 14), "small": (20, 128, 1, 12), "deep": (38, 256, 3, 12)}.items():
        for seed in (0, 1):
            sd = orc.make_ff_state_dict(I, H, n_hidden, O, seed)
            model = ref_nn.DropoutFF(output_size=O, hidden_layer_size=H, hidden_layer_count=n_hidden, input_size=I, dropout=0.2)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            model.eval()
            rng = np.random.default_rng(200 + seed)
            for shape in ((1, 6, I), (37, 6, I), (300, I)):
                x = rng.normal(size=shape).astype(np.float32)
                with torch.no_grad():
                    y = model(torch.from_numpy(x)).numpy()
                key = f"{tag}_seed{seed}_" + "x".join(map(str, shape))
                blob["x_" + key] = x
                blob["y_" + key] = y
        blob["dims_" + tag] = np.array([I, H, n_hidden, O])
    np.savez_compressed(OUT / "ff.npz", **blob)


# ---- 3. quaternion primitive goldens ------------------------------------------------------
def gen_imupose():
    """ImuPoseLSTM (nn_models.py:210-249), the third architecture the loader dispatches (:397-398), eval mode"""
    blob = {}
    for tag, (I, O) in {"pocket_like": (22, 14), "uarm_like": (38, 12)}.items():
        for seed in (0, 1):
            sd = orc.make_imupose_state_dict(I, O, seed)
            model = ref_nn.ImuPoseLSTM(input_size=I, hidden_layer_size=128, hidden_layer_count=3, output_size=O, dropout=0.2)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            model.eval()
            blob[f"digest_{tag}_seed{seed}"] = orc.state_dict_digest(sd)
            rng = np.random.default_rng(300 + seed)
            for (B, T) in ((1, 6), (21, 6), (3, 64), (2, 1)):
                x = rng.normal(size=(B, T, I)).astype(np.float32)
                with torch.no_grad():
                    y = model(torch.from_numpy(x)).numpy()
                    ymc = model.monte_carlo_predictions(5, torch.from_numpy(x[:1])).numpy()
                blob[f"x_{tag}_seed{seed}_B{B}_T{T}"] = x
                blob[f"y_{tag}_seed{seed}_B{B}_T{T}"] = y
                blob[f"ymc_{tag}_seed{seed}_B{B}_T{T}"] = ymc       # [1,T,O]: no repeat, no dropout (:246-251)
        blob["dims_" + tag] = np.array([I, O])
    np.savez_compressed(OUT / "imupose.npz", **blob)


def edge_six_drr(rng, n):
    """6D rows that stress the Gram-Schmidt / quaternion branches."""
    rows = []
    # exact rotation matrices of random quaternions, incl. rotations by ~pi (w ~ 0)
    q = rand_unit_quats(rng, n)
    q[: n // 4, 0] *= 1e-9
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r9 = ref_ts.quat_to_rot_mat_1x9(q)
    rows.append(r9[:, [0, 1, 3, 4, 6, 7]])
    # un-normalised, non-orthogonal network-like outputs
    rows.append(rng.normal(size=(n, 6)))
    # nearly parallel columns
    a = rng.normal(size=(n, 3))
    b = a * rng.uniform(0.5, 2.0, size=(n, 1)) + 1e-4 * rng.normal(size=(n, 3))
    rows.append(np.stack([a[:, 0], b[:, 0], a[:, 1], b[:, 1], a[:, 2], b[:, 2]], axis=1))
    # axis-aligned: identity and 180-degree flips (trace = -1 pivots)
    for diag in ((1, 1, 1), (1, -1, -1), (-1, 1, -1), (-1, -1, 1)):
        rows.append(np.array([[diag[0], 0, 0, diag[1], 0, 0]], dtype=np.float64))
    rows.append(np.array([[0, 1, 0, 0, 1, 0.0], [0, 0, 1, 0, 0, 1.0], [0, -1, 1, 0, 0, 0.0]]))
    return np.vstack(rows)


def gen_quat_ops():
    rng = np.random.default_rng(7)
    six = edge_six_drr(rng, 40)
    r9 = ref_ts.six_drr_1x6_to_rot_mat_1x9(six)
    quat = ref_ts.rot_mat_1x9_to_quat(r9)
    a, b = rand_unit_quats(rng, 50), rng.normal(size=(50, 4))
    v = rng.normal(size=(50, 3))
    sn, cs = rng.normal(size=60), rng.normal(size=60)
    sn[:6] = [0.0, 0.0, 1e-300, -1e-12, 0.0, 1.0]
    cs[:6] = [0.0, -1.0, 1e-300, -1.0, 1.0, 0.0]
    avg_sets = []
    for n in (2, 5, 60, 300):
        base = rand_unit_quats(rng, 1)
        qs = base + 0.2 * rng.normal(size=(n, 4))
        qs /= np.linalg.norm(qs, axis=1, keepdims=True)
        qs[1::3] *= -1.0          # antipodal duplicates exercise the sign alignment
        avg_sets.append(qs)
    blob = dict(
        six=six, rotmat=r9, quat=quat,
        ham_a=a, ham_b=b, ham=ref_ts.hamilton_product(a, b),
        rot_q=a, rot_v=v, rot_out=ref_ts.quat_rotate_vector(a, v),
        rot_single_v=v[0], rot_single_out=ref_ts.quat_rotate_vector(a, v[0]),
        hips_sin=sn, hips_cos=cs, hips_quat=ref_ts.hips_sin_cos_to_quat(sn, cs),
    )
    for i, qs in enumerate(avg_sets):
        blob[f"avg_in_{i}"] = qs
        blob[f"avg_out_{i}"] = ref_ts.average_quaternions(qs)
    np.savez_compressed(OUT / "quat_ops.npz", **blob)


# ---- 4. FK + message goldens ------------------------------------------------------------
def gen_fk(stats):
    rng = np.random.default_rng(11)
    body_default = orc.DEFAULT_BODY
    body_other = np.array([[-0.25, 0.0, 0.0, -0.3, 0.0, 0.0, -0.18, 0.45, 0.01]])
    for tname, layout in LAYOUT_OF.items():
        O = orc.LAYOUT_NUM_TARGETS[layout]
        tgt = NNS_TARGETS[tname]
        blob = {}
        for tag, body in (("bd", body_default), ("bo", body_other)):
            for N in (1, 7, 300):
                if layout == orc.LAYOUT_ORI_CAL_LARM_UARM_HIPS:
                    st = stats["pocket"]
                    preds = np.array(st["yy_m"]) + 3.0 * np.array(st["yy_s"]) * rng.uniform(-1, 1, size=(N, O))
                elif layout == orc.LAYOUT_ORI_CAL_LARM_UARM:
                    st = stats["watch"]
                    preds = np.array(st["yy_m"]) + 3.0 * np.array(st["yy_s"]) * rng.uniform(-1, 1, size=(N, O))
                else:
                    preds = rng.normal(size=(N, O))
                if N == 300:   # splice edge-case 6D rows and degenerate hips into the big set
                    e = edge_six_drr(rng, 20)
                    lo, uo = {0: (0, 6), 1: (0, 6), 2: (3, 12)}[layout]
                    preds[: len(e), lo:lo + 6] = e
                    preds[: len(e), uo:uo + 6] = e[::-1]
                    if layout != orc.LAYOUT_ORI_CAL_LARM_UARM:
                        preds[0, -2:] = 0.0
                        preds[1, -2:] = [0.0, -1.0]
                        preds[2, -2:] = [1e-200, 1e-200]
                est = ref_fk.arm_pose_from_nn_targets(preds, body, tgt)
                msg = ref_msg.msg_from_nn_targets_est(est, body, tgt)
                blob[f"preds_{tag}_N{N}"] = preds
                blob[f"est_{tag}_N{N}"] = est
                blob[f"msg_{tag}_N{N}"] = np.asarray(msg, dtype=np.float64)
            blob[f"body_{tag}"] = body
        np.savez_compressed(OUT / f"fk_layout{layout}.npz", **blob)


# ---- 5. streaming traces through the reference Estimator subclasses ----------------------
def synth_rows(rng, n_frames, width, lookup):
    rows = rng.normal(size=(n_frames, width)).astype(np.float32)
    rows[:, lookup["sw_dt"]] = 0.02
    for pre in ("sw_rotvec", "sw_forward", "ph_rotvec", "ph_forward"):
        if pre + "_w" in lookup:
            idx = [lookup[f"{pre}_{c}"] for c in "wxyz"]
            if pre.endswith("forward"):   # calibration quaternion: constant over a recording
                rows[:, idx] = rand_unit_quats(rng, 1).astype(np.float32)
            else:                          # smooth-ish random walk on the sphere
                q = rand_unit_quats(rng, 1)
                out = []
                for _ in range(n_frames):
                    q = q + 0.05 * rng.normal(size=(1, 4))
                    q /= np.linalg.norm(q)
                    out.append(q[0])
                rows[:, idx] = np.array(out, dtype=np.float32)
    rows[:, lookup["sw_pres"]] = 1000.0 + rng.normal(size=n_frames).astype(np.float32)
    rows[:, lookup["sw_init_pres"]] = 1000.5
    return rows


def gen_stream_traces():
    from wear_mocap_ape.data_types import messaging
    from wear_mocap_ape.estimate.watch_only import WatchOnlyNN
    from wear_mocap_ape.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    from wear_mocap_ape.estimate.watch_phone_uarm_nn import WatchPhoneUarmNN

    classes = {
        "pocket": (WatchPhonePocketNN, messaging.WATCH_PHONE_IMU_LOOKUP),
        "watch": (WatchOnlyNN, messaging.WATCH_ONLY_IMU_LOOKUP),
        "uarm": (WatchPhoneUarmNN, messaging.WATCH_PHONE_IMU_LOOKUP),
    }
    seed = 3
    real_loader = ref_nn.load_deployed_model_from_hash

    for name, (cls, lookup) in classes.items():
        def fake_load(hash_str, _name=name):
            # checkpoints are absent: same class + params as nn_models.py:390-408, seeded weights,
            # dropout=0 so that the permanent lstm.train() of nn_models.py:204 stays deterministic
            model, p, _ = ref_model(_name, seed, dropout=0.0)
            return model, p

        ref_nn.load_deployed_model_from_hash = fake_load
        rng = np.random.default_rng(21)
        rows = synth_rows(rng, 20, len(lookup), lookup)
        blob = {"rows": rows, "weights_seed": np.array(seed)}
        for smooth, mc in ((1, 1), (5, 1), (3, 4)):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                est = cls(model_hash=HASHES[name], smooth=smooth, add_mc_samples=True, monte_carlo_samples=mc)
            xs, preds, msgs = [], [], []
            for row32 in rows:
                row = array("f", row32.tolist())     # wire type of ImuListener (stream/listener/imu.py:66-69)
                xx = est.parse_row_to_xx(row)
                pred = est.add_xx_to_row_hist_and_make_prediction(xx)
                msg = est.msg_from_pred(pred, True)
                xs.append(np.asarray(xx, dtype=np.float64))
                preds.append(pred)
                msgs.append(np.asarray(msg, dtype=np.float64))
            tag = f"s{smooth}_mc{mc}"
            blob[f"xx_{tag}"] = np.array(xs)
            blob[f"xx_dtype_{tag}"] = np.array(str(np.asarray(est.parse_row_to_xx(array("f", rows[0].tolist()))).dtype))
            blob[f"pred_{tag}"] = np.array(preds)
            blob[f"msg_{tag}"] = np.array(msgs)
            blob[f"last_msg_{tag}"] = np.asarray(est.get_last_msg(), dtype=np.float64)
            blob["body"] = est.body_measurements
            blob["seq_len"] = np.array(est.sequence_len)
        np.savez_compressed(OUT / f"stream_trace_{name}.npz", **blob)
    ref_nn.load_deployed_model_from_hash = real_loader


def edge_forward_quats(rng):
    """android-frame calibration quaternions whose azimuth (transformations.py:200-207) covers the circle and its corners: the global y
    rotation by `a` is the android quaternion (-cos(a/2), 0, 0, sin(a/2)) (android_quat_to_global_no_north, :225-233)"""
    ang = [0.0, np.pi, -np.pi, np.pi / 2, -np.pi / 2, np.pi - 1e-6, -np.pi + 1e-6, 1e-7, -1e-7, 3.0, -3.0, 0.75 * np.pi, -0.75 * np.pi, 1e-3]
    ang += list(np.linspace(-np.pi, np.pi, 25))
    q = [np.array([-np.cos(0.5 * a), 0.0, 0.0, np.sin(0.5 * a)]) for a in ang]
    q += [np.array([0.0, 0.0, 0.0, 1.0]), np.array([0.0, 0.0, 0.0, -1.0]), np.array([-1.0, 0.0, 0.0, 0.0]), np.array([1.0, 0.0, 0.0, 0.0])]
    r = rand_unit_quats(rng, 30)
    q += list(r[:16])                                           # tilted calibration poses
    q += list(r[16:20] * 1e-18) + list(r[20:24] * 1e15)         # un-normalised: the azimuth does not depend on the norm
    q += list(r[24:30] * rng.uniform(0.2, 3.0, size=(6, 1)))
    q += [np.zeros(4)]                                          # no calibration at all: atan2(0, 0)
    return np.array(q)


def gen_feature_edges():
    """parse_row_to_xx of the three reference estimators on rows whose calibration quaternions sweep the azimuths (the recorded traces of
    gen_stream_traces hold ONE forward quaternion each): fixture for the feature builder's angle arithmetic"""
    from wear_mocap_ape.data_types import messaging
    from wear_mocap_ape.estimate.watch_only import WatchOnlyNN
    from wear_mocap_ape.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    from wear_mocap_ape.estimate.watch_phone_uarm_nn import WatchPhoneUarmNN
    classes = {
        "pocket": (WatchPhonePocketNN, messaging.WATCH_PHONE_IMU_LOOKUP),
        "watch": (WatchOnlyNN, messaging.WATCH_ONLY_IMU_LOOKUP),
        "uarm": (WatchPhoneUarmNN, messaging.WATCH_PHONE_IMU_LOOKUP),
    }
    real_loader = ref_nn.load_deployed_model_from_hash
    blob = {}
    for name, (cls, lookup) in classes.items():
        def fake_load(hash_str, _name=name):
            model, p, _ = ref_model(_name, 3, dropout=0.0)
            return model, p
        ref_nn.load_deployed_model_from_hash = fake_load
        rng = np.random.default_rng(33)
        fwd = edge_forward_quats(rng)
        rows = synth_rows(rng, len(fwd), len(lookup), lookup)
        rows[:, [lookup[f"sw_forward_{c}"] for c in "wxyz"]] = fwd.astype(np.float32)
        if "ph_forward_w" in lookup:     # the phone's own calibration: the same sweep, in another order; three rows with rotation == calibration
            ph = fwd[rng.permutation(len(fwd))].astype(np.float32)
            rows[:, [lookup[f"ph_forward_{c}"] for c in "wxyz"]] = ph
            rows[:3, [lookup[f"ph_rotvec_{c}"] for c in "wxyz"]] = ph[:3]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            est = cls(model_hash=HASHES[name], smooth=1, add_mc_samples=True, monte_carlo_samples=1)
            xs = []
            with np.errstate(all="ignore"):
                for row32 in rows:
                    xs.append(np.asarray(est.parse_row_to_xx(array("f", row32.tolist())), dtype=np.float64))
        blob[f"rows_{name}"] = rows
        blob[f"xx_{name}"] = np.array(xs)
    ref_nn.load_deployed_model_from_hash = real_loader
    np.savez_compressed(OUT / "feature_edges.npz", **blob)


def gen_trace_mc_stats():
    """The reference ESTIMATORS in their Monte-Carlo mode, end to end: each class is built with its deployed dropout rate and
    4000 samples per frame, driven through the 20-row trace of `stream_trace_<name>.npz` exactly as `processing_loop` does
    (estimator.py:174-177), and the hand / elbow rows `msg_from_pred` appends to the message (estimator.py:131-137) are kept for
    the LAST frame; six passes (reset in between, as `processing_loop` resets on start) give 24 000 samples of one window's
    distribution.  Mean, covariance and quantiles go to trace_mc_stats.npz: the drop-in loop of the build (one
    `process_row` per message, device-resident) is held to them by tests/mc_check.py."""
    from wear_mocap_ape.estimate.watch_only import WatchOnlyNN
    from wear_mocap_ape.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    from wear_mocap_ape.estimate.watch_phone_uarm_nn import WatchPhoneUarmNN
    classes = {"pocket": WatchPhonePocketNN, "watch": WatchOnlyNN, "uarm": WatchPhoneUarmNN}
    real_loader = ref_nn.load_deployed_model_from_hash
    per_frame = 4000
    blob = {"quantile_levels": np.array(MC_QUANTILES), "n_samples": np.array(MC_SAMPLES), "weights_seed": np.array(0)}
    for name, cls in classes.items():
        def fake_load(hash_str, _name=name):
            model, p, _ = ref_model(_name, 0)            # deployed dropout rate (results.json)
            return model, p
        ref_nn.load_deployed_model_from_hash = fake_load
        rows = np.load(OUT / f"stream_trace_{name}.npz")["rows"]
        torch.manual_seed(4242)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            est = cls(model_hash=HASHES[name], smooth=1, add_mc_samples=True, monte_carlo_samples=per_frame)
        tails, means = [], []
        for _ in range(MC_SAMPLES // per_frame):
            est.reset()
            for row32 in rows:
                row = array("f", row32.tolist())
                msg = est.msg_from_pred(est.add_xx_to_row_hist_and_make_prediction(est.parse_row_to_xx(row)), True)
            msg = np.asarray(msg, dtype=np.float64)
            assert msg.shape == (25 + 6 * per_frame,)
            tails.append(msg[25:].reshape(per_frame, 6))
            means.append(msg[:25])
        t6 = np.concatenate(tails)
        blob[f"est6_mean_{name}"] = t6.mean(axis=0)
        blob[f"est6_cov_{name}"] = np.cov(t6, rowvar=False)
        blob[f"est6_quant_{name}"] = np.quantile(t6, MC_QUANTILES, axis=0)
        blob[f"msg_mean_{name}"] = np.array(means)        # the six 4000-sample mean-pose messages (sign-aligned quaternion means)
        blob[f"dropout_{name}"] = np.array(ref_params(name)["dropout"])
    ref_nn.load_deployed_model_from_hash = real_loader
    np.savez_compressed(OUT / "trace_mc_stats.npz", **blob)


def gen_bookkeeping():
    """column-name enums and UDP message lookups, verbatim values from the reference (data)"""
    from wear_mocap_ape.data_types import messaging
    from wear_mocap_ape.data_types.bone_map import BoneMap
    names = {
        "NNS_INPUTS": {m.name: list(m.value) for m in NNS_INPUTS},
        "NNS_TARGETS": {m.name: m.value for m in NNS_TARGETS if isinstance(m.value, list)},
        "WATCH_ONLY_IMU_LOOKUP": dict(messaging.WATCH_ONLY_IMU_LOOKUP),
        "WATCH_PHONE_IMU_LOOKUP": dict(messaging.WATCH_PHONE_IMU_LOOKUP),
        "watch_only_imu_msg_len": messaging.watch_only_imu_msg_len,
        "watch_phone_imu_msg_len": messaging.watch_phone_imu_msg_len,
        "deploy_hashes": {m.name: m.value for m in ref_deploy.LSTM},
        "bone_defaults": {"larm": BoneMap.DEFAULT_LARM_LEN, "uarm": BoneMap.DEFAULT_UARM_LEN,
                          "uarm_orig_rh": BoneMap.DEFAULT_UARM_ORIG_RH.tolist()},
    }
    (OUT / "bookkeeping.json").write_text(json.dumps(names, indent=1))


def gen_csv_header():
    """the header line the reference's recorder writes (record/est_output.py:16-32), as data"""
    import tempfile
    from wear_mocap_ape.record.est_output import EstOutputRecorder
    with tempfile.TemporaryDirectory() as d:
        f = Path(d) / "est.csv"
        EstOutputRecorder(f)
        header = f.read_text().strip().split(",")
    (OUT / "est_csv_header.json").write_text(json.dumps(header))


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    if sys.argv[1:] == ["csv"]:              # add this one fixture without rewriting the others
        gen_csv_header()
        print("wrote", OUT / "est_csv_header.json")
        return
    if sys.argv[1:] == ["hs"]:               # add these fixtures without rewriting the others
        gen_lstm_hs()
        print("wrote", OUT / "lstm_hs.npz")
        return
    if sys.argv[1:] == ["trace_mc"]:         # add this one fixture without rewriting the others
        gen_trace_mc_stats()
        print("wrote", OUT / "trace_mc_stats.npz")
        return
    if sys.argv[1:] == ["feature_edges"]:    # add this one fixture without rewriting the others
        gen_feature_edges()
        print("wrote", OUT / "feature_edges.npz")
        return
    if sys.argv[1:] == ["mc"]:
        gen_mc_stats()
        print("wrote", OUT / "mc_stats.npz")
        return
    if sys.argv[1:] == ["imupose"]:          # add this one fixture without rewriting the others
        gen_imupose()
        print("wrote", OUT / "imupose.npz")
        return
    stats = export_stats_and_configs()
    gen_bookkeeping()
    gen_lstm(stats)
    gen_lstm_hs()
    gen_mc_stats()
    gen_ff()
    gen_imupose()
    gen_quat_ops()
    gen_fk(stats)
    gen_stream_traces()
    gen_trace_mc_stats()
    gen_feature_edges()
    gen_csv_header()
    total = sum(f.stat().st_size for f in OUT.glob("*.np*")) + (OUT / "norm_stats.json").stat().st_size
    print("golden fixtures written to", OUT, f"({total / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
