"""Round-6 GPU tests (through the C ABI of libape_hip.so): user-built one-layer LSTMs of the widths the one-layer cluster FORM exists for
(ADVICE r05), and the level-synchronous kernel behind the drop-in surface."""
import numpy as np
import pytest
import torch

from oracle import ape_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as entry
    entry.build()


@pytest.mark.parametrize("I,H,O", [(22, 256, 14), (32, 256, 12), (38, 128, 12), (64, 128, 6)])
def test_one_layer_models_of_the_layer0_widths_are_created_and_run(I, H, O):
    """`hidden_layer_count=1` is a valid reference configuration (nn_models.py:161-178).  Round 5 let `ape_cluster_supported` answer true for
    the one-layer FORM the Monte-Carlo banks launch for layer 0 (H = 256 / KX = 32, H = 128 / KX = 64), and `ape_model_create` took the whole
    cluster set-up for such a MODEL -- whose fp16 and multi-tile instantiations do not exist: 'cluster kernel set-up failed'.  The full-model
    predicate is back to full models; these run on the batch-tile kernel, against the oracle (1e-6), eval and all-steps."""
    from wear_mocap_ape_amd.estimate import nn_models
    sd = orc.make_state_dict(I, H, 1, O, 4)
    m = nn_models.DropoutLSTM(I, H, 1, O, dropout=0.2, device=0)
    m.load_state_dict(sd)
    rng = np.random.default_rng(I + H)
    for B, T in ((7, 6), (600, 6), (33, 20)):
        x = rng.normal(size=(B, T, I)).astype(np.float32)
        y = m(torch.from_numpy(x).cuda(), last_step_only=True).cpu().numpy()[:, 0]
        assert "tile16" in m.last_kernel(), m.last_kernel()
        m.check()
        ref = orc.lstm_forward(sd, x)
        assert np.abs(y - ref[:, -1]).max() < 1e-6
        ya = m(torch.from_numpy(x).cuda(), last_step_only=False).cpu().numpy()
        m.check()
        assert ya.shape == (B, T, O) and np.abs(ya - ref).max() < 1e-6


def test_level16_serves_the_upper_arm_estimators_eval_windows(norm_stats):
    """the route a caller of `Estimator.infer_windows` / the eval bank takes at the deployed shape: 1024, 512 and 40 windows x 6 steps of the 3 x 128 model go
    to `ape_lstm_level16` under AUTO (two row tiles per cluster / one), 1025 do not, `cluster_gen1` switches it off with the other second-generation kernels -- and all
    of them agree with the oracle; two calls on two model handles interleaved on one stream keep their launch numbers apart (the tags of
    one handle's granules mean nothing to the other's buffer)."""
    from tests.test_hip_parity import make_model, _synthetic_windows
    st = norm_stats["uarm"]
    m, sd, cfg = make_model("uarm", 9, st)
    m2, sd2, _ = make_model("uarm", 10, st)
    for B, want in ((1024, "ape_lstm_level16"), (512, "ape_lstm_level16"), (40, "ape_lstm_level16"), (1025, "ape_lstm_cluster")):
        x = _synthetic_windows(st, B, 6, cfg["I"], B)
        xn = ((x.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        ys = []
        for rep in range(3):                                  # interleaved: handle 1, handle 2, handle 1 ..
            ys.append(m(xd, last_step_only=True, normalize_input=True))
            assert m.last_kernel() == want, (B, m.last_kernel())
            ys.append(m2(xd, last_step_only=True, normalize_input=True))
        m.check(); m2.check()
        r1, r2 = orc.lstm_forward(sd, xn)[:, -1], orc.lstm_forward(sd2, xn)[:, -1]
        for i, y in enumerate(ys):
            assert np.abs(y.cpu().numpy()[:, 0] - (r1 if i % 2 == 0 else r2)).max() < 1e-6
    m.set_kernel("cluster_gen1")
    x = _synthetic_windows(st, 1024, 6, cfg["I"], 3)
    m(torch.from_numpy(x).cuda(), last_step_only=True, normalize_input=True)
    assert m.last_kernel() == "ape_lstm_cluster"
    m.check()


@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_feature_builder_over_the_azimuths(golden, tmp_path, monkeypatch, name):
    """`ape_parse_rows` on `feature_edges.npz` -- the reference's own parse_row_to_xx outputs on 74 rows whose calibration quaternions sweep
    the azimuth circle, its corners (0, +-pi, +-pi/2, pi - 1e-6, +-1e-7), tilted and un-normalised poses (1e-18 .. 1e15) and the zero
    quaternion.  Round 6: the builder gets cos / sin of the azimuth and of half of it from half-angle identities instead of
    atan2 -> cos / sin (csrc/angle_device.h); the recorded traces hold ONE calibration quaternion each, this fixture holds the circle.
    float32 rows against the reference at 1e-6 (2e-6 upper arm: part of ITS quaternion math is float32), float64
    rows against the host builder (the reference's formulas in float64; itself held to this fixture on the CPU) at 1e-12; NaN exactly
    where the reference has NaN."""
    from array import array
    from tests.test_hip_parity import _deploy_dir
    from wear_mocap_ape_amd import config
    from wear_mocap_ape_amd.estimate.watch_only import WatchOnlyNN
    from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    from wear_mocap_ape_amd.estimate.watch_phone_uarm_nn import WatchPhoneUarmNN
    g = golden("feature_edges.npz")
    deploy, h = _deploy_dir(tmp_path, name, 3, dropout=0.0)
    monkeypatch.setitem(config.PATHS, "deploy", deploy)
    est = {"pocket": WatchPhonePocketNN, "watch": WatchOnlyNN, "uarm": WatchPhoneUarmNN}[name](model_hash=h)
    rows, ref = g[f"rows_{name}"], g[f"xx_{name}"]
    xx = est.parse_rows(rows).cpu().numpy().astype(np.float64)
    assert xx.shape == ref.shape and np.array_equal(np.isnan(xx), np.isnan(ref))
    assert np.nanmax(np.abs(xx - ref)) < (2e-6 if name == "uarm" else 1e-6), np.nanmax(np.abs(xx - ref))
    xx64 = est.parse_rows(rows, out_dtype=torch.float64).cpu().numpy()
    with np.errstate(all="ignore"):
        host = np.array([np.asarray(est.parse_row_to_xx(array("f", r.tolist())), dtype=np.float64) for r in rows])
    assert np.array_equal(np.isnan(xx64), np.isnan(host))
    assert np.nanmax(np.abs(xx64 - host)) < (1e-12 if name == "uarm" else 1e-6), np.nanmax(np.abs(xx64 - host))


@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_feature_builder_on_random_messages(tmp_path, monkeypatch, name):
    """20 000 random messages per estimator (every column standard normal: un-normalised rotation and calibration quaternions, azimuths all
    round the circle, both branches of the half-angle pair) through `ape_parse_rows` in float64 against the host builder (the reference's
    formulas in float64, itself held to the reference's outputs on `feature_edges.npz` and the recorded traces): 1e-12 for the upper-arm
    estimator, whose host builder returns float64; 2e-6 for the two whose builders return float32 like the reference's."""
    from array import array
    from tests.test_hip_parity import _deploy_dir
    from wear_mocap_ape_amd import config, _hip
    from wear_mocap_ape_amd.estimate.watch_only import WatchOnlyNN
    from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    from wear_mocap_ape_amd.estimate.watch_phone_uarm_nn import WatchPhoneUarmNN
    deploy, h = _deploy_dir(tmp_path, name, 3, dropout=0.0)
    monkeypatch.setitem(config.PATHS, "deploy", deploy)
    est = {"pocket": WatchPhonePocketNN, "watch": WatchOnlyNN, "uarm": WatchPhoneUarmNN}[name](model_hash=h)
    kind = {"pocket": _hip.PARSE_WATCH_PHONE_POCKET, "watch": _hip.PARSE_WATCH_ONLY, "uarm": _hip.PARSE_WATCH_PHONE_UARM}[name]
    rng = np.random.default_rng(77)
    rows = rng.normal(size=(20000, _hip.PARSE_SHAPES[kind][0])).astype(np.float32)
    rows[::7] *= rng.uniform(1e-3, 1e3, size=(len(rows[::7]), 1)).astype(np.float32)
    xx64 = est.parse_rows(rows, out_dtype=torch.float64).cpu().numpy()
    host = np.array([np.asarray(est.parse_row_to_xx(array("f", r.tolist())), dtype=np.float64) for r in rows])
    assert np.isfinite(xx64).all() and np.isfinite(host).all()
    # (sensor columns are copied: exact; the 6D rotation and the yaw features are the arithmetic under test)
    err = np.abs(xx64 - host) / np.maximum(1.0, np.abs(host))
    assert err.max() < (1e-12 if name == "uarm" else 2e-6), (err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.mark.parametrize("layout,O", [(0, 14), (2, 20)])
def test_post_filter_on_random_targets_all_round_the_hips_circle(layout, O):
    """`ape_fk` on 100 000 random prediction rows per hips-carrying target layout against the oracle (float64, 1e-11): the hips quaternion comes
    from half-angle identities since round 6 (csrc/angle_device.h) and the goldens hold a few hundred rows; the (sin, cos) pairs here cover the
    circle, tiny and large magnitudes included."""
    from wear_mocap_ape_amd.estimate import _post
    rng = np.random.default_rng(layout + 5)
    N = 100000
    preds = rng.normal(size=(N, O))
    preds[::11, -2:] *= rng.uniform(1e-6, 1e6, size=(len(preds[::11]), 1))
    preds[::13, -1] = -np.abs(preds[::13, -1])                 # cos < 0: the other branch of the half-angle pair
    preds[::17, -2] *= 1e-9                                    # azimuths next to 0 and to +-pi
    ctx = _post.context(layout)
    est = _post.fk_rows(ctx.handle, layout, ctx.device, preds, orc.DEFAULT_BODY)
    ref = orc.arm_pose_from_targets(preds, orc.DEFAULT_BODY, layout, "closed")
    assert est.shape == ref.shape and np.isfinite(est).all()
    assert np.abs(est - ref).max() < 1e-11, np.abs(est - ref).max()
