"""GPU parity tests: the HIP path (through the C ABI of libape_hip.so) against
  * the golden vectors produced by the reference itself (tests/golden/),
  * the pinned CPU oracle on seeded inputs,
  * size-independent properties at BASELINE.json's full sizes.

Tolerances.  BASELINE.json / SURVEY.md 8d state float32 budgets of 1e-5 (T <= 8) and 1e-4 (T = 64)
on the NN targets and 5e-5 on quaternions/origins.  The f32-MFMA kernel measures ~5e-8 (first GPU
run: max|dy| 4.3e-8, max|dquat| 7.2e-8 at B=1024, T=64), so the tests assert 20x tighter bounds:
  NN targets  <= 1e-6 abs (any T)
  est rows    <= 1e-11 abs for the float64 post-filter alone; <= 2e-6 end to end
  bookkeeping (row/column indices, lengths, duplicated/constant message fields): bit-exact
"""
import ctypes as C
import json
import shutil
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import ape_oracle as orc

pytestmark = pytest.mark.gpu

TOL_Y_SHORT = 1e-6
TOL_Y_T64 = 1e-6
TOL_FK64 = 1e-11
TOL_EST_E2E = 2e-6


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as entry
    entry.build()


def make_model(name, seed, stats=None):
    from wear_mocap_ape_amd.estimate import nn_models
    cfg = orc.MODEL_CONFIGS[name]
    sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], seed)
    m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], dropout=0.2, device=0)
    m.load_state_dict(sd)
    if stats is not None:
        m.set_norm_stats(stats["xx_m"], stats["xx_s"], stats["yy_m"], stats["yy_s"])
    return m, sd, cfg


def quat_err(a, b, amb=1e-4):
    """strict where |w_ref| is clear of zero, sign-aware only where the reference w ~ 0"""
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    dp, dm = np.abs(a - b).max(axis=1), np.abs(a + b).max(axis=1)
    return float(np.nanmax(np.where(np.abs(b[:, 0]) < amb, np.minimum(dp, dm), dp)))


# ---------------- LSTM + head vs the reference's own outputs -------------------------------------
@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_lstm_vs_reference_golden(golden, name):
    g = golden(f"lstm_{name}.npz")
    for seed in (0, 1):
        model, sd, cfg = make_model(name, seed)
        assert np.array_equal(orc.state_dict_digest(sd), g[f"digest_seed{seed}"])
        for (B, T) in ((1, cfg["T"]), (5, cfg["T"]), (3, 64), (2, 1)):
            x, y_ref = g[f"x_seed{seed}_B{B}_T{T}"], g[f"y_seed{seed}_B{B}_T{T}"]
            y = model(torch.from_numpy(x))                       # host in -> host out, all steps
            assert tuple(y.shape) == (B, T, cfg["O"]) and y.dtype == torch.float32 and not y.is_cuda
            err = float(np.abs(y.numpy() - y_ref).max())
            assert err < (TOL_Y_T64 if T > 8 else TOL_Y_SHORT), (name, seed, B, T, err)
            # device in -> device out, last step only: AUTO picks the cluster kernel (other summation order)
            y_last = model(torch.from_numpy(x).cuda(), last_step_only=True)
            assert y_last.is_cuda and tuple(y_last.shape) == (B, 1, cfg["O"])
            assert np.abs(y_last.cpu().numpy()[:, 0] - y_ref[:, -1]).max() < TOL_Y_SHORT
            # within the batch-tile kernel the last-step-only and all-steps outputs are the same bits
            model.set_kernel("tile16")
            y16_all = model(torch.from_numpy(x).cuda())
            y_t16 = model(torch.from_numpy(x).cuda(), last_step_only=True)
            assert np.array_equal(y_t16.cpu().numpy()[:, 0], y16_all.cpu().numpy()[:, -1])
            # AUTO's all-steps route (cluster kernel + head over the [B,T,H] rows) against it, every step
            assert np.abs(y.numpy() - y16_all.cpu().numpy()).max() < TOL_Y_SHORT
            model.set_kernel("auto")


@pytest.mark.parametrize("name,B,T", [("pocket", 17, 6), ("pocket", 60, 6), ("watch", 33, 8), ("uarm", 16, 6),
                                      ("uarm", 1, 3), ("pocket", 1, 1), ("watch", 130, 2)])
def test_lstm_vs_oracle_ragged_batches(name, B, T):
    """batch sizes that leave partial 16-row tiles, windows shorter than the deployed ones"""
    model, sd, cfg = make_model(name, 7)
    x = np.random.default_rng(B * 100 + T).normal(size=(B, T, cfg["I"])).astype(np.float32)
    y = model(torch.from_numpy(x)).numpy()
    y_ref = orc.lstm_forward(sd, x)
    assert np.abs(y - y_ref).max() < TOL_Y_SHORT


@pytest.mark.parametrize("name,B,T", [("pocket", 1, 6), ("pocket", 60, 6), ("pocket", 17, 1), ("pocket", 300, 6),
                                      ("pocket", 1024, 8), ("pocket", 1100, 3), ("watch", 1, 8), ("watch", 65, 8),
                                      ("uarm", 1, 6), ("uarm", 50, 6), ("uarm", 700, 4), ("uarm", 2100, 2)])
def test_cluster_kernel_vs_oracle_and_tile16(name, B, T):
    """the weight-stationary cluster kernel (all row-tile counts, partial clusters, more than one
    launch per call) against the oracle and against the batch-tile kernel"""
    model, sd, cfg = make_model(name, 11)
    x = np.random.default_rng(B * 7 + T).normal(size=(B, T, cfg["I"])).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    y_cl = model.set_kernel("cluster")(xt, last_step_only=True).cpu().numpy()[:, 0]
    model.check()
    y_t16 = model.set_kernel("tile16")(xt, last_step_only=True).cpu().numpy()[:, 0]
    y_ref = orc.lstm_forward(sd, x)[:, -1]
    assert np.abs(y_t16 - y_ref).max() < TOL_Y_SHORT
    assert np.abs(y_cl - y_ref).max() < TOL_Y_SHORT, float(np.abs(y_cl - y_ref).max())
    # repeatable: same launch twice gives the same bits (flags re-zeroed, no stale exchange data)
    y_again = model.set_kernel("cluster")(xt, last_step_only=True).cpu().numpy()[:, 0]
    assert np.array_equal(y_cl, y_again)
    model.check()


def test_mlp_regressor_vs_reference_golden(golden):
    """DropoutFF (the MLP regressor of nn_models.py:313-370) against the reference module's outputs"""
    from wear_mocap_ape_amd.estimate import nn_models
    g = golden("ff.npz")
    for tag in ("pocket_like", "small", "deep"):
        I, H, n_hidden, O = (int(v) for v in g["dims_" + tag])
        for seed in (0, 1):
            sd = orc.make_ff_state_dict(I, H, n_hidden, O, seed)
            m = nn_models.DropoutFF(output_size=O, hidden_layer_size=H, hidden_layer_count=n_hidden, input_size=I, device=0)
            m.load_state_dict(sd)
            assert list(m.state_dict().keys()) == orc.ff_state_dict_keys(n_hidden)
            for shape in ((1, 6, I), (37, 6, I), (300, I)):
                key = f"{tag}_seed{seed}_" + "x".join(map(str, shape))
                y = m(torch.from_numpy(g["x_" + key])).numpy()
                assert y.shape == g["y_" + key].shape
                assert np.abs(y - g["y_" + key]).max() < TOL_Y_SHORT, (key, float(np.abs(y - g["y_" + key]).max()))
            # last step only == the last row of the all-steps output (same arithmetic)
            x = g[f"x_{tag}_seed{seed}_37x6x{I}"]
            assert np.array_equal(m(torch.from_numpy(x), last_step_only=True).numpy()[:, 0], m(torch.from_numpy(x)).numpy()[:, -1])
            # injected dropout mask vs the oracle
            rng = np.random.default_rng(1)
            mask = (rng.uniform(size=(37, H)) >= 0.2).astype(np.float32) / 0.8
            ym = m(torch.from_numpy(x), masks=torch.from_numpy(mask), last_step_only=True).numpy()[:, 0]
            assert np.abs(ym - orc.ff_forward(sd, x[:, -1], mask=mask)).max() < TOL_Y_SHORT
            # MC mode: dropout in front of the output layer, permanent, seeded, batch rows repeated
            m.manual_seed(5)
            a = m.monte_carlo_predictions(50, torch.from_numpy(x[:1]), last_step_only=True).numpy()[:, 0]
            m.manual_seed(5)
            b = m.monte_carlo_predictions(50, torch.from_numpy(x[:1]), last_step_only=True).numpy()[:, 0]
            assert a.shape == (50, O) and np.array_equal(a, b) and a.std(axis=0).min() > 1e-5 and m._do.training


def test_imupose_lstm_vs_reference_golden(golden, norm_stats, tmp_path, monkeypatch):
    """ImuPoseLSTM (nn_models.py:210-249: Linear+ReLU, fixed 2 x 256 LSTM, Linear) against the reference module's
    outputs; loader dispatch; fused normalisation; stream bank on top of it"""
    from wear_mocap_ape_amd import config
    from wear_mocap_ape_amd.estimate import nn_models
    g = golden("imupose.npz")
    for tag in ("pocket_like", "uarm_like"):
        I, O = (int(v) for v in g["dims_" + tag])
        for seed in (0, 1):
            sd = orc.make_imupose_state_dict(I, O, seed)
            m = nn_models.ImuPoseLSTM(I, 128, 3, O, device=0)             # size arguments kept but ignored (:217-229)
            m.load_state_dict(sd)
            assert list(m.state_dict().keys()) == orc.imupose_state_dict_keys()
            for (B, T) in ((1, 6), (21, 6), (3, 64), (2, 1)):
                key = f"{tag}_seed{seed}_B{B}_T{T}"
                x = torch.from_numpy(g["x_" + key])
                y = m(x).numpy()
                assert y.shape == g["y_" + key].shape == (B, T, O)
                assert np.abs(y - g["y_" + key]).max() < TOL_Y_SHORT, (key, float(np.abs(y - g["y_" + key]).max()))
                # (on the cluster kernel the all-steps output takes the head in a second launch: another summation order)
                assert np.abs(m(x, last_step_only=True).numpy()[:, 0] - y[:, -1]).max() < 1e-6
                ymc = m.monte_carlo_predictions(5, x[:1]).numpy()          # plain forward: no repeat, no dropout
                assert ymc.shape == (1, T, O) and np.abs(ymc - g["ymc_" + key]).max() < TOL_Y_SHORT
            m.check()
    # a ragged bigger batch vs the oracle, raw features with the z-score fused in front of the input layer
    stats = norm_stats["pocket"]
    sd = orc.make_imupose_state_dict(22, 14, 7)
    m = nn_models.ImuPoseLSTM(22, 256, 2, 14, device=0)
    m.load_state_dict(sd)
    m.set_norm_stats(stats["xx_m"], stats["xx_s"], stats["yy_m"], stats["yy_s"])
    raw = _synthetic_windows(stats, 77, 6, 22, 3)
    xn = ((raw.astype(np.float64) - stats["xx_m"]) / stats["xx_s"]).astype(np.float32)
    y = m(torch.from_numpy(raw), normalize_input=True, last_step_only=True).numpy()[:, 0]
    assert np.abs(y - orc.imupose_forward(sd, xn)[:, -1]).max() < TOL_Y_SHORT
    with pytest.raises(UserWarning):
        m(torch.from_numpy(raw), masks=torch.zeros((1, 77, 6, 256)))
    # loader dispatch (nn_models.py:397-398)
    d = tmp_path / "nn" / "imuhash"
    d.mkdir(parents=True)
    (d / "results.json").write_text(json.dumps({"model": "ImuPoseLSTM", "hidden_layer_size": 128, "hidden_layer_count": 3,
                                                "dropout": 0.2, "x_inputs_v": ["f"] * 22, "y_targets_v": ["t"] * 14}))
    torch.save(({k: torch.from_numpy(v) for k, v in sd.items()}, {}), d / "checkpoint.pt")
    monkeypatch.setitem(config.PATHS, "deploy", tmp_path)
    model, params = nn_models.load_deployed_model_from_hash("imuhash")
    assert isinstance(model, nn_models.ImuPoseLSTM) and params["model"] is nn_models.ImuPoseLSTM
    assert np.abs(model(torch.from_numpy(xn)).numpy() - orc.imupose_forward(sd, xn)).max() < TOL_Y_SHORT


def test_loader_dispatches_dropout_ff(tmp_path, monkeypatch):
    """results.json with "model": "DropoutFF" + a (model_state, optimizer_state) checkpoint -> HIP MLP"""
    from wear_mocap_ape_amd import config
    from wear_mocap_ape_amd.estimate import nn_models
    I, H, n_hidden, O = 22, 256, 2, 14
    d = tmp_path / "nn" / "ffhash"
    d.mkdir(parents=True)
    (d / "results.json").write_text(json.dumps({"model": "DropoutFF", "hidden_layer_size": H, "hidden_layer_count": n_hidden,
                                                "dropout": 0.2, "x_inputs_v": ["f"] * I, "y_targets_v": ["t"] * O}))
    sd = orc.make_ff_state_dict(I, H, n_hidden, O, 4)
    torch.save(({k: torch.from_numpy(v) for k, v in sd.items()}, {}), d / "checkpoint.pt")
    monkeypatch.setitem(config.PATHS, "deploy", tmp_path)
    model, params = nn_models.load_deployed_model_from_hash("ffhash")
    assert isinstance(model, nn_models.DropoutFF) and params["model"] is nn_models.DropoutFF
    x = np.random.default_rng(0).normal(size=(9, 6, I)).astype(np.float32)
    assert np.abs(model(torch.from_numpy(x)).numpy() - orc.ff_forward(sd, x)).max() < TOL_Y_SHORT


@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_small_batch_latency_kernel(norm_stats, name):
    """B <= 4: the VALU/shuffle variant of the cluster kernel (AUTO) against the oracle and the MFMA kernels"""
    st = norm_stats[name]
    model, sd, cfg = make_model(name, 12, st)
    for B in (1, 2, 3, 4):
        for T in (1, cfg["T"], 64):
            x = _synthetic_windows(st, B, T, cfg["I"], 10 * B + T)
            xn = ((x.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
            y_ref = orc.lstm_forward(sd, xn)[:, -1]
            xt = torch.from_numpy(x).cuda()
            y_small = model.set_kernel("auto")(xt, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
            y_mfma = model.set_kernel("cluster")(xt, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
            assert np.abs(y_small - y_ref).max() < TOL_Y_SHORT, (B, T, float(np.abs(y_small - y_ref).max()))
            assert np.abs(y_mfma - y_ref).max() < TOL_Y_SHORT
            again = model.set_kernel("auto")(xt, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
            assert np.array_equal(again, y_small)                  # self-cleaning state, deterministic
            # the kernel exchanges inside one XCD's L2 when its members turn out to share an XCD, write-through
            # otherwise; the internal flag forces the second form: same arithmetic, same bits
            from wear_mocap_ape_amd import _hip
            y_wt = torch.empty((B, cfg["O"]), dtype=torch.float32, device="cuda")
            _hip.check(_hip.lib().ape_lstm_forward(model.handle, C.c_void_p(xt.data_ptr()), B, T,
                                                   _hip.FLAG_NORMALIZE_INPUT | 0x08000000, None, 0.0, 0,
                                                   C.c_void_p(y_wt.data_ptr()), None), "ape_lstm_forward")
            torch.cuda.synchronize()
            assert np.array_equal(y_wt.cpu().numpy(), y_small)
            # on a device whose XCDs hold H/8 CUs the kernel runs with H/8 members (two units per wave); the internal flag
            # forces the H/16-member form: same arithmetic up to the f32 summation order
            y_w4 = torch.empty((B, cfg["O"]), dtype=torch.float32, device="cuda")
            _hip.check(_hip.lib().ape_lstm_forward(model.handle, C.c_void_p(xt.data_ptr()), B, T,
                                                   _hip.FLAG_NORMALIZE_INPUT | 0x01000000, None, 0.0, 0,
                                                   C.c_void_p(y_w4.data_ptr()), None), "ape_lstm_forward")
            torch.cuda.synchronize()
            assert np.abs(y_w4.cpu().numpy() - y_ref).max() < TOL_Y_SHORT
    assert "cluster" in model.kernel_name(1, 6)
    model.check()


def test_fused_normalisation_is_bit_exact(norm_stats):
    """APE_FLAG_NORMALIZE_INPUT == host float64 z-score followed by the float32 cast"""
    st = norm_stats["pocket"]
    model, sd, cfg = make_model("pocket", 2, st)
    rng = np.random.default_rng(5)
    x_raw = (st["xx_m"] + st["xx_s"] * rng.normal(size=(40, 6, cfg["I"]))).astype(np.float32)
    x_norm = ((x_raw.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
    y_fused = model(torch.from_numpy(x_raw), normalize_input=True).numpy()
    y_host = model(torch.from_numpy(x_norm)).numpy()
    assert np.array_equal(y_fused, y_host)


@pytest.mark.parametrize("name", ["pocket", "uarm"])
def test_dropout_with_injected_masks(name):
    """MC-dropout parity with explicit masks: torch's internal mask stream cannot be replayed
    (SURVEY.md 3.3), so the oracle's cell loop and the kernels get the same masks"""
    model, sd, cfg = make_model(name, 3)
    B, T, p = 25, cfg["T"], 0.2
    rng = np.random.default_rng(9)
    x = np.repeat(rng.normal(size=(1, T, cfg["I"])).astype(np.float32), B, axis=0)
    masks = (rng.uniform(size=(cfg["L"] - 1, B, T, cfg["H"])) >= p).astype(np.float32) / (1 - p)
    y_ref = orc.lstm_forward(sd, x, masks=list(masks))
    y = model(torch.from_numpy(x), masks=torch.from_numpy(masks)).numpy()        # all steps: cluster kernel + head rows
    assert np.abs(y - y_ref).max() < TOL_Y_SHORT
    y16 = model.set_kernel("tile16")(torch.from_numpy(x), masks=torch.from_numpy(masks)).numpy()   # all steps, batch-tile kernel
    model.set_kernel("auto")
    assert np.abs(y16 - y_ref).max() < TOL_Y_SHORT
    assert np.abs(y[0] - y[1]).max() > 1e-4          # masks differ per row -> rows differ
    for kern in ("tile16", "cluster"):               # last step only: both kernels
        yk = model.set_kernel(kern)(torch.from_numpy(x), masks=torch.from_numpy(masks), last_step_only=True).numpy()
        assert np.abs(yk[:, 0] - y_ref[:, -1]).max() < TOL_Y_SHORT, kern
    model.check()


@pytest.mark.parametrize("name,n", [("pocket", 60), ("pocket", 400), ("uarm", 50), ("watch", 25)])
def test_philox_dropout_same_masks_in_both_kernels(name, n):
    """the cluster kernel and the batch-tile kernel draw the same Philox masks for the same seed"""
    model, sd, cfg = make_model(name, 6)
    xt = torch.from_numpy(np.random.default_rng(n).normal(size=(1, cfg["T"], cfg["I"])).astype(np.float32))
    out = {}
    for kern in ("tile16", "cluster"):
        model.set_kernel(kern)
        model.manual_seed(123)
        out[kern] = model.monte_carlo_predictions(n, xt, last_step_only=True).numpy()[:, 0]
    assert out["cluster"].std(axis=0).min() > 1e-4
    assert np.abs(out["cluster"] - out["tile16"]).max() < TOL_Y_SHORT
    model.check()


def test_philox_dropout_statistics():
    """in-kernel masks: deterministic per seed, different across calls, unbiased on average"""
    model, sd, cfg = make_model("pocket", 4)
    x = np.random.default_rng(3).normal(size=(1, 6, cfg["I"])).astype(np.float32)
    xt = torch.from_numpy(x)
    y_eval = model(xt).numpy()[0, -1]
    model.manual_seed(11)
    a = model.monte_carlo_predictions(400, xt).numpy()[:, -1]
    b = model.monte_carlo_predictions(400, xt).numpy()[:, -1]
    model.manual_seed(11)
    a2 = model.monte_carlo_predictions(400, xt).numpy()[:, -1]
    assert np.array_equal(a, a2) and not np.array_equal(a, b)
    assert model.lstm.training                                   # permanent, like nn_models.py:204
    assert a.std(axis=0).min() > 1e-4                            # samples really differ
    # inverted dropout keeps the layer-1 input unbiased: the MC mean stays near the eval output
    ref_spread = np.abs(a - y_eval).mean()
    assert np.abs(a.mean(axis=0) - y_eval).max() < 0.35 * max(ref_spread, 1e-3) + 0.02
    # p = 0 switches dropout off entirely
    model.dropout = 0.0
    assert np.array_equal(model.monte_carlo_predictions(3, xt).numpy()[0, -1], y_eval)
    with pytest.raises(UserWarning, match="batch size 1"):
        model.monte_carlo_predictions(3, torch.zeros(2, 6, cfg["I"]))


# ---------------- FK + message vs the reference's own outputs ---------------------------------------
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_fk_and_msg_vs_reference_golden(golden, layout):
    from wear_mocap_ape_amd.estimate import compose_msg, estimate_joints
    from wear_mocap_ape_amd.utility.names import NNS_TARGETS
    tgt = {0: NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS, 1: NNS_TARGETS.ORI_CAL_LARM_UARM,
           2: NNS_TARGETS.ORI_POS_CAL_LARM_UARM_HIPS}[layout]
    g = golden(f"fk_layout{layout}.npz")
    W = orc.LAYOUT_EST_WIDTH[layout]
    qcols = {0: (9, 13, 17), 1: (6, 10), 2: (9, 13, 17)}[layout]
    for tag in ("bd", "bo"):
        body = g[f"body_{tag}"]
        for N in (1, 7, 300):
            preds, est_ref, msg_ref = g[f"preds_{tag}_N{N}"], g[f"est_{tag}_N{N}"], g[f"msg_{tag}_N{N}"]
            est = estimate_joints.arm_pose_from_nn_targets(preds, body, tgt)
            assert est.shape == (N, W) and est.dtype == np.float64
            good = ~np.isnan(est_ref).any(axis=1)
            assert np.array_equal(good, ~np.isnan(est).any(axis=1))       # degenerate rows stay NaN
            for c in qcols:
                assert quat_err(est[good, c:c + 4], est_ref[good, c:c + 4]) < TOL_FK64
            clear = good & (np.abs(est_ref[:, list(qcols)]) > 1e-4).all(axis=1)
            assert np.abs(est[clear, :qcols[0]] - est_ref[clear, :qcols[0]]).max() < TOL_FK64
            # message from the reference's est rows: isolates averaging + layout bookkeeping
            rows = est_ref if (N == 1 or good.all()) else est_ref[good]
            msg = compose_msg.msg_from_nn_targets_est(rows, body, tgt)
            assert msg.shape == (25,) and msg.dtype == np.float64
            if N == 1 or good.all():
                assert np.abs(msg - msg_ref).max() < TOL_FK64
            assert np.array_equal(msg[0:4], msg[7:11])                     # hand rot == larm rot (bit-exact)
            if layout == 1:
                assert np.array_equal(msg[21:25], [1.0, 0.0, 0.0, 0.0])    # constant hips
                assert np.array_equal(msg[18:21], body[0, 6:9])            # constant upper-arm origin


def test_fk_storage_types():
    """f32/f64 preds and est buffers through ape_fk: same float64 arithmetic, rounded once"""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.estimate import _post
    rng = np.random.default_rng(2)
    preds = rng.normal(size=(513, 14))
    ctx = _post.context(0)
    ref = _post.fk_rows(ctx.handle, 0, ctx.device, preds, orc.DEFAULT_BODY)
    assert np.abs(ref - orc.arm_pose_from_targets(preds, orc.DEFAULT_BODY, 0, "closed")).max() < TOL_FK64
    p32 = torch.from_numpy(preds.astype(np.float32)).cuda()
    e32 = torch.empty((513, 21), dtype=torch.float32, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _hip.check(_hip.lib().ape_fk(ctx.handle, C.c_void_p(p32.data_ptr()), _hip.F32, 513, 0,
                                 C.c_void_p(e32.data_ptr()), _hip.F32, stream), "ape_fk")
    ref32 = orc.arm_pose_from_targets(preds.astype(np.float32).astype(np.float64), orc.DEFAULT_BODY, 0, "closed")
    assert np.array_equal(e32.cpu().numpy(), ref32.astype(np.float32)) or \
        np.abs(e32.cpu().numpy() - ref32).max() < 2e-7


def test_error_behaviour():
    from wear_mocap_ape_amd.estimate import estimate_joints
    from wear_mocap_ape_amd.utility.names import NNS_TARGETS
    model, sd, cfg = make_model("pocket", 0)
    with pytest.raises(UserWarning):
        model(torch.zeros(0, 6, cfg["I"]))                       # empty batch
    with pytest.raises(UserWarning):
        model(torch.zeros(2, 0, cfg["I"]))                       # empty window
    with pytest.raises(UserWarning):
        model(torch.zeros(2, 6, cfg["I"] + 1))                   # wrong feature count
    with pytest.raises(UserWarning):
        model(torch.zeros(2, 6, cfg["I"]), hs=(1, 2))            # carried state is not part of the path
    with pytest.raises(UserWarning):
        estimate_joints.arm_pose_from_nn_targets(np.zeros((3, 12)), orc.DEFAULT_BODY, NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS)
    with pytest.raises(RuntimeError):
        model.load_state_dict({"lstm.weight_ih_l0": np.zeros((4, 4))})
    # C-ABI status codes of the stream bank (they reach Python as UserWarning through _hip.check)
    from wear_mocap_ape_amd import _hip
    lib = _hip.lib()
    model.set_body(orc.DEFAULT_BODY)
    h = C.c_void_p()
    assert lib.ape_streams_create(model.handle, 0, 6, 1, C.byref(h)) != 0            # no streams
    assert lib.ape_streams_create(model.handle, 4, 6, 65, C.byref(h)) != 0           # smooth beyond one wave
    assert lib.ape_streams_create(model.handle, 4, 6, 2, C.byref(h)) == 0
    assert lib.ape_streams_set_mc(h, 0, 0.2, 1) != 0                                 # no samples
    assert lib.ape_streams_set_mc(h, 4096, 0.2, 1) != 0                              # smooth * n_mc beyond 4096
    assert lib.ape_streams_set_mc(h, 3, 1.0, 1) != 0                                 # dropout_p must stay below 1
    assert lib.ape_streams_set_mc(h, 3, 0.2, 1) == 0
    msg = torch.empty((4, 25 + 6 * 6), dtype=torch.float32, device="cuda")
    assert lib.ape_streams_step(h, 0, C.c_void_p(msg.data_ptr()), None, _hip.F32, None) != 0      # nothing pushed yet
    xx = torch.zeros((4, cfg["I"]), dtype=torch.float32, device="cuda")
    assert lib.ape_streams_push_features(h, C.c_void_p(xx.data_ptr()), None) == 0
    assert lib.ape_streams_push_rows(h, _hip.PARSE_WATCH_ONLY, C.c_void_p(xx.data_ptr()), None) != 0   # 20 features, model takes 22
    assert lib.ape_streams_step(h, _hip.FLAG_PACKED_MSG, C.c_void_p(msg.data_ptr()), None, 7, None) != 0          # unknown dtype selector
    assert lib.ape_streams_step(h, _hip.FLAG_PACKED_MSG, C.c_void_p(msg.data_ptr()), C.c_void_p(msg.data_ptr()), _hip.F32, None) != 0
    assert lib.ape_streams_step(h, _hip.FLAG_ALL_STEPS, C.c_void_p(msg.data_ptr()), None, _hip.F32, None) != 0     # not a step flag
    assert lib.ape_streams_step(h, _hip.FLAG_PACKED_MSG, C.c_void_p(msg.data_ptr()), None, _hip.F32, None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(msg).all()
    assert len(lib.ape_last_error()) > 0                                             # the last failure left its text
    assert lib.ape_streams_destroy(h) == 0


# ---------------- the estimator classes: streaming traces of the reference ----------------------------
def _deploy_dir(tmp_path, name, seed, dropout):
    """a deploy tree with a synthetic checkpoint in the reference's (model_state, optimizer_state)
    tuple format (nn_models.py:410)"""
    from wear_mocap_ape_amd import config
    cfg = orc.MODEL_CONFIGS[name]
    src = Path(config.PATHS["deploy"])
    dst = tmp_path / "deploy"
    shutil.copytree(src / "data_stats", dst / "data_stats", dirs_exist_ok=True)
    hashes = {"pocket": "670b66fa7664252d1cfb3b5a8a362002ffeeba5c", "watch": "04f4ad63bfccb3668f7598c9375403e10b1fae2a",
              "uarm": "7cb5cdf94ef4c66388c7f15f642005d5e008146a"}
    d = dst / "nn" / hashes[name]
    d.mkdir(parents=True, exist_ok=True)
    p = json.loads((src / "nn" / hashes[name] / "results.json").read_text())
    p["dropout"] = dropout
    (d / "results.json").write_text(json.dumps(p))
    sd = orc.make_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], seed)
    torch.save(({k: torch.from_numpy(v) for k, v in sd.items()}, {"state": {}, "param_groups": []}), d / "checkpoint.pt")
    return dst, hashes[name]


@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_estimator_stream_trace(golden, tmp_path, monkeypatch, name):
    from array import array
    from wear_mocap_ape_amd import config
    from wear_mocap_ape_amd.estimate.watch_only import WatchOnlyNN
    from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    from wear_mocap_ape_amd.estimate.watch_phone_uarm_nn import WatchPhoneUarmNN
    g = golden(f"stream_trace_{name}.npz")
    deploy, h = _deploy_dir(tmp_path, name, int(g["weights_seed"]), dropout=0.0)   # as the golden run
    monkeypatch.setitem(config.PATHS, "deploy", deploy)
    cls = {"pocket": WatchPhonePocketNN, "watch": WatchOnlyNN, "uarm": WatchPhoneUarmNN}[name]
    cfg = orc.MODEL_CONFIGS[name]
    for smooth, mc in ((1, 1), (5, 1), (3, 4)):
        tag = f"s{smooth}_mc{mc}"
        est = cls(model_hash=h, smooth=smooth, add_mc_samples=True, monte_carlo_samples=mc)
        assert est.sequence_len == cfg["T"]
        assert np.array_equal(est.body_measurements, g["body"])
        worst_pred = worst_msg = 0.0
        for f, row32 in enumerate(g["rows"]):
            xx = est.parse_row_to_xx(array("f", row32.tolist()))
            pred = est.add_xx_to_row_hist_and_make_prediction(xx)
            pred_ref = g[f"pred_{tag}"][f]
            assert pred.shape == pred_ref.shape and pred.dtype == np.float64      # bookkeeping: rows = smooth*mc
            worst_pred = max(worst_pred, float(np.abs(pred - pred_ref).max()))
            msg = est.msg_from_pred(pred_ref, True)          # reference preds in -> isolates FK + message + MC tail
            msg_ref = g[f"msg_{tag}"][f]
            n_rows = pred_ref.shape[0]
            assert isinstance(msg, list) and len(msg) == len(msg_ref) == (25 + 6 * n_rows if n_rows > 1 else 25)
            worst_msg = max(worst_msg, float(np.abs(np.asarray(msg) - msg_ref).max()))
        assert worst_pred < 2e-6, (tag, worst_pred)          # de-normalised targets (yy_s <= 1)
        assert worst_msg < 1e-10, (tag, worst_msg)
        assert np.abs(est.get_last_msg() - g[f"last_msg_{tag}"]).max() < 1e-10 and len(est.get_last_msg()) == 25
        est.reset()
        assert est._row_hist == [] and est._smooth_hist == [] and not est.is_active()


@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_batched_feature_builder(golden, tmp_path, monkeypatch, name):
    """ape_parse_rows against the reference's own parse_row_to_xx outputs"""
    from wear_mocap_ape_amd import config
    from wear_mocap_ape_amd.estimate.watch_only import WatchOnlyNN
    from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    from wear_mocap_ape_amd.estimate.watch_phone_uarm_nn import WatchPhoneUarmNN
    g = golden(f"stream_trace_{name}.npz")
    deploy, h = _deploy_dir(tmp_path, name, 3, dropout=0.0)
    monkeypatch.setitem(config.PATHS, "deploy", deploy)
    est = {"pocket": WatchPhonePocketNN, "watch": WatchOnlyNN, "uarm": WatchPhoneUarmNN}[name](model_hash=h)
    ref = g["xx_s1_mc1"]
    xx = est.parse_rows(g["rows"])
    assert xx.is_cuda and tuple(xx.shape) == ref.shape
    assert str(xx.dtype).endswith(str(g["xx_dtype_s1_mc1"]))          # float32 / float64 like the reference
    # uarm: the reference's own quaternion math is partly float32 there
    tol = 2e-6 if name == "uarm" else 1e-6
    assert np.abs(xx.cpu().numpy().astype(np.float64) - ref).max() < tol
    # float64 output vs the host feature builder (same formulas in float64): tight
    from array import array
    xx64 = est.parse_rows(g["rows"], out_dtype=torch.float64).cpu().numpy()
    host = np.array([np.asarray(est.parse_row_to_xx(array("f", r.tolist())), dtype=np.float64) for r in g["rows"]])
    assert np.abs(xx64 - host).max() < (1e-12 if name == "uarm" else 1e-6)
    # a big ragged batch: rows are independent, so tiling the trace must tile the output
    rows = np.tile(g["rows"], (53, 1))[:1001]
    big = est.parse_rows(rows).cpu().numpy()
    assert np.array_equal(big, np.tile(xx.cpu().numpy(), (53, 1))[:1001])
    if name == "watch":      # the watch-only estimator fed by the 55-float watch+phone message
        est2 = WatchOnlyNN(model_hash=h, watch_phone=True)
        rows55 = golden("stream_trace_pocket.npz")["rows"]
        out = est2.parse_rows(rows55).cpu().numpy()
        host2 = np.array([est2.parse_row_to_xx(array("f", r.tolist())) for r in rows55])
        assert out.shape == (len(rows55), 20) and np.abs(out - host2).max() < 1e-6
    with pytest.raises(UserWarning):
        est.parse_rows(np.zeros((3, 7), np.float32))


def test_processing_loop_thread(tmp_path, monkeypatch, golden):
    """the consumer thread contract: sensor_q in -> msg_q out, terminate() stops it"""
    import queue
    import time
    from array import array
    from wear_mocap_ape_amd import config
    from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    g = golden("stream_trace_pocket.npz")
    deploy, h = _deploy_dir(tmp_path, "pocket", 3, dropout=0.2)
    monkeypatch.setitem(config.PATHS, "deploy", deploy)
    est = WatchPhonePocketNN(model_hash=h, smooth=2, monte_carlo_samples=10)
    sensor_q = queue.Queue()
    msg_q = est.process_in_thread(sensor_q)
    try:
        for row32 in g["rows"][:4]:
            sensor_q.put(array("f", row32.tolist()))
            msg = msg_q.get(timeout=30)
            assert len(msg) == 25 + 6 * 20
            q = np.asarray(msg[0:4])
            assert abs(np.linalg.norm(q) - 1.0) < 1e-9
        assert est.is_active()
    finally:
        est.terminate()
        time.sleep(0.1)


# ---------------- the batched path at BASELINE.json's full sizes --------------------------------------
def _synthetic_windows(st, B, T, I, seed):
    rng = np.random.default_rng(seed)
    x = st["xx_m"] + st["xx_s"] * rng.normal(size=(B, T, I))
    x[..., 0] = 0.02                                    # 50 Hz
    return x.astype(np.float32)


@pytest.mark.parametrize("name", ["pocket", "watch"])
def test_full_size_batched_path(norm_stats, name):
    """config 3 / 5 shapes: 1024 windows x 64 frames, against the oracle and through properties"""
    from wear_mocap_ape_amd import _hip
    st = norm_stats[name]
    model, sd, cfg = make_model(name, 0, st)
    model.set_body(orc.DEFAULT_BODY)
    B, T, W = 1024, 64, orc.LAYOUT_EST_WIDTH[cfg["layout"]]
    x = _synthetic_windows(st, B, T, cfg["I"], 1)
    xd = torch.from_numpy(x).cuda()
    y = torch.empty((B, cfg["O"]), dtype=torch.float32, device="cuda")
    est = torch.empty((B, W), dtype=torch.float64, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(inp, yo, eo):
        _hip.check(_hip.lib().ape_infer(model.handle, C.c_void_p(inp.data_ptr()), inp.shape[0], T,
                                        _hip.FLAG_NORMALIZE_INPUT, C.c_void_p(yo.data_ptr()),
                                        C.c_void_p(eo.data_ptr()), _hip.F64, stream), "ape_infer")
        torch.cuda.synchronize()

    run(xd, y, est)
    y_ref, est_ref = orc.infer_windows(sd, st, orc.DEFAULT_BODY, cfg["layout"], x)
    ey = float(np.abs(y.cpu().numpy() - y_ref).max())
    qcols = (9, 13, 17) if cfg["layout"] == 0 else (6, 10)
    e = est.cpu().numpy()
    eq = max(quat_err(e[:, c:c + 4], est_ref[:, c:c + 4]) for c in qcols)
    eo = float(np.abs(e[:, :qcols[0]] - est_ref[:, :qcols[0]]).max())
    print(f"\n[{name} B={B} T={T}] max|dy|={ey:.2e} max|dquat|={eq:.2e} max|dorig|={eo:.2e}")
    assert ey < TOL_Y_T64 and eq < TOL_EST_E2E and eo < TOL_EST_E2E

    # windows are independent: a permutation of the batch permutes the output bit-exactly
    perm = torch.from_numpy(np.random.default_rng(0).permutation(B)).cuda()
    y2, est2 = torch.empty_like(y), torch.empty_like(est)
    run(xd[perm].contiguous(), y2, est2)
    assert torch.equal(y2, y[perm]) and torch.equal(est2, est[perm])
    # ... and so does a ragged sub-batch (partial last tile)
    y3, est3 = torch.empty((1001, cfg["O"]), dtype=torch.float32, device="cuda"), torch.empty((1001, W), dtype=torch.float64, device="cuda")
    run(xd[:1001].contiguous(), y3, est3)
    assert torch.equal(y3, y[:1001]) and torch.equal(est3, est[:1001])
    # unit quaternions with w >= 0; bone lengths preserved by the kinematic chain
    for c in qcols:
        q = e[:, c:c + 4]
        assert np.abs(np.linalg.norm(q, axis=1) - 1.0).max() < 1e-12 and (q[:, 0] >= 0).all()
    assert np.abs(np.linalg.norm(e[:, 0:3] - e[:, 3:6], axis=1) - 0.22).max() < 1e-12
    uo = e[:, 6:9] if cfg["layout"] == 0 else orc.DEFAULT_BODY[:, 6:9]
    assert np.abs(np.linalg.norm(e[:, 3:6] - uo, axis=1) - 0.26).max() < 1e-12


@pytest.mark.parametrize("name,B,T", [("watch", 1024, 64), ("watch", 1, 8), ("watch", 60, 8), ("pocket", 1024, 64),
                                      ("pocket", 77, 6), ("uarm", 300, 6)])
def test_fp16_hidden_state_variant(norm_stats, name, B, T):
    """BASELINE.json configs[4]: watch-only model, batch 1024, fp16 hidden state / weights with fp32
    accumulate.  Two checks: (1) against the float32 oracle within the STATED tolerance of the config
    (5e-3 abs on the NN targets; measured value printed); (2) against the oracle's binary16-storage
    emulation, which pins layout/indexing far below that tolerance."""
    st = norm_stats[name]
    model, sd, cfg = make_model(name, 0, st)
    x = _synthetic_windows(st, B, T, cfg["I"], 2)
    xn = ((x.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
    y16 = model.set_precision("f16")(torch.from_numpy(x).cuda(), last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    model.check()
    y32 = model.set_precision("f32")(torch.from_numpy(x).cuda(), last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    y_ref = orc.lstm_forward(sd, xn)[:, -1]
    y_emu = orc.lstm_forward(sd, xn, storage="f16")[:, -1]
    e_ref, e_emu = float(np.abs(y16 - y_ref).max()), float(np.abs(y16 - y_emu).max())
    print(f"\n[{name} B={B} T={T} fp16] max|y16 - f32 oracle|={e_ref:.2e}  max|y16 - f16-emulating oracle|={e_emu:.2e}  "
          f"(f32 kernel vs oracle {np.abs(y32 - y_ref).max():.1e})")
    assert np.abs(y32 - y_ref).max() < TOL_Y_T64
    assert e_ref < 5e-3                       # stated tolerance of configs[4]
    assert e_emu < 3e-4                       # same arithmetic up to f32 summation order and rare 1-ulp f16 flips
    with pytest.raises(UserWarning):          # fp16 variant: last step only, no dropout
        model.set_precision("f16")(torch.from_numpy(x[:1]).cuda())
    model.set_precision("f32")


def test_graph_capture_and_replay(norm_stats):
    """ape_infer is capturable into a hipGraph (kernels only -- the cluster kernel cleans its own flags --, no allocation once reserved) and a
    replay on new input data reproduces the eager result bit for bit"""
    from wear_mocap_ape_amd import _hip
    st = norm_stats["pocket"]
    model, sd, cfg = make_model("pocket", 8, st)
    model.set_body(orc.DEFAULT_BODY)
    lib = _hip.lib()
    for B, T in ((1, 6), (200, 6)):
        _hip.check(lib.ape_model_reserve(model.handle, B), "reserve")
        x = torch.from_numpy(_synthetic_windows(st, B, T, cfg["I"], 3)).cuda()
        x2 = torch.from_numpy(_synthetic_windows(st, B, T, cfg["I"], 4)).cuda()
        est_eager = torch.empty((B, 21), dtype=torch.float64, device="cuda")
        est_graph = torch.zeros_like(est_eager)
        xin = x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            st_ptr = C.c_void_p(side.cuda_stream)
            call = lambda out: _hip.check(lib.ape_infer(model.handle, C.c_void_p(xin.data_ptr()), B, T,
                                                        _hip.FLAG_NORMALIZE_INPUT, None, C.c_void_p(out.data_ptr()),
                                                        _hip.F64, st_ptr), "ape_infer")
            call(est_graph)                        # warm-up outside capture
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                call(est_graph)
        torch.cuda.current_stream().wait_stream(side)
        for data in (x, x2, x):
            xin.copy_(data)
            graph.replay()
            torch.cuda.synchronize()
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _hip.check(lib.ape_infer(model.handle, C.c_void_p(data.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT, None,
                                     C.c_void_p(est_eager.data_ptr()), _hip.F64, stream), "ape_infer")
            torch.cuda.synchronize()
            assert torch.equal(est_graph, est_eager)
        model.check()


def test_infer_windows_entry(tmp_path, monkeypatch, norm_stats):
    """Estimator.infer_windows == per-window streaming through the same estimator"""
    from wear_mocap_ape_amd import config
    from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    deploy, h = _deploy_dir(tmp_path, "pocket", 5, dropout=0.0)
    monkeypatch.setitem(config.PATHS, "deploy", deploy)
    est = WatchPhonePocketNN(model_hash=h, smooth=1, monte_carlo_samples=1)
    x = _synthetic_windows(norm_stats["pocket"], 9, 6, 22, 4)
    rows = est.infer_windows(x).cpu().numpy()
    assert rows.shape == (9, 21)
    for b in range(9):
        est.reset()
        for t in range(6):
            pred = est.add_xx_to_row_hist_and_make_prediction(x[b, t])
        # after 6 pushes the window holds exactly x[b]; single row -> message copies the est row
        msg = np.asarray(est.msg_from_pred(pred, False))
        assert np.abs(msg[4:7] - rows[b, 0:3]).max() < 1e-6
        assert quat_err(msg[7:11], rows[b, 9:13]) < 1e-6


# ---------------- stream bank: window rings + smoothing stacks + messages on the device ---------------------
@pytest.mark.parametrize("name,S,kernel", [("pocket", 7, "auto"), ("pocket", 3, "auto"), ("pocket", 21, "tile16"),
                                           ("watch", 9, "auto"), ("uarm", 6, "auto")])
def test_stream_bank(golden, norm_stats, name, S, kernel):
    """ape_streams_* (SURVEY 8a-1, 8a-15, 8f-2): S streams stepped together against (i) the reference's own
    end-to-end trace on stream 0 and (ii) the oracle's window/smoothing bookkeeping + FK + message on all of them"""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    g = golden(f"stream_trace_{name}.npz")
    stats, body = norm_stats[name], g["body"]
    m, sd, cfg = make_model(name, int(g["weights_seed"]), stats)
    m.set_body(body)
    if kernel == "tile16":
        m.set_kernel("tile16")
    kind = {"pocket": _hip.PARSE_WATCH_PHONE_POCKET, "watch": _hip.PARSE_WATCH_ONLY, "uarm": _hip.PARSE_WATCH_PHONE_UARM}[name]
    rows = g["rows"].astype(np.float32)
    F, T = len(rows), cfg["T"]
    rng = np.random.default_rng(5)
    order = [np.arange(F)] + [rng.permutation(F) for _ in range(S - 1)]       # stream 0 replays the golden trace
    # features per raw row, float32 like the reference's parse_row_to_xx (the builder has its own test)
    xx_all = torch.empty((F, cfg["I"]), dtype=torch.float32, device="cuda")
    _hip.check(_hip.lib().ape_parse_rows(kind, C.c_void_p(torch.from_numpy(rows).cuda().data_ptr()), F,
                                         C.c_void_p(xx_all.data_ptr()), _hip.F32, None), "ape_parse_rows")
    torch.cuda.synchronize()
    xx_all = xx_all.cpu().numpy()
    for smooth in (1, 5):
        bank = StreamBank(m, S, T, smooth=smooth, normalize=True, dtype=torch.float64)
        predict = lambda hist: orc.lstm_forward(sd, np.asarray(hist, dtype=np.float32)[None])[:, -1, :]
        wins = [orc.WindowOracle(T, smooth, stats, predict) for _ in range(S)]
        worst_ref = worst_orc = worst_tail = 0.0
        for rnd in range(2):                                  # second round after reset(): cold start again
            for f in range(F):
                batch = np.stack([rows[order[s][f]] for s in range(S)])
                if f % 2:       # every other frame arrives as the UDP payload: big-endian float32
                    dev = torch.from_numpy(batch.astype(">f4").view(np.float32)).cuda()
                    bank.push_rows(dev, kind, big_endian=True)
                else:
                    bank.push_rows(torch.from_numpy(batch).cuda(), kind)
                if rnd == 1 and smooth > 1:     # second round: the UDP payload layout (message + tail in one float32 row)
                    d = bank.step_datagrams().cpu().numpy()
                    assert d.dtype == np.float32 and d.shape == (S, 25 + 6 * smooth)
                    assert len(d[0].tobytes()) == 4 * len(g[f"msg_s{smooth}_mc1"][f])      # pose_est_udp.py:47
                    msg, tail = d[:, :25].astype(np.float64), d[:, 25:].reshape(S, smooth, 6).astype(np.float64)
                else:
                    msg, tail = bank.step(with_tail=True)
                    msg, tail = msg.cpu().numpy(), tail.cpu().numpy()
                assert msg.shape == (S, 25) and tail.shape == (S, smooth, 6)
                for s in range(S):
                    pred = wins[s].push(xx_all[order[s][f]])
                    assert pred.shape == (smooth, cfg["O"])
                    est = orc.arm_pose_from_targets(pred, body, cfg["layout"], "eigh")
                    ref = orc.msg_from_est(est, body, cfg["layout"])
                    worst_orc = max(worst_orc, float(np.abs(msg[s] - ref).max()))
                    worst_tail = max(worst_tail, float(np.abs(tail[s] - est[:, :6]).max()))
                    assert np.array_equal(msg[s, 0:4], msg[s, 7:11])          # hand rot duplicates larm rot
                    if cfg["layout"] == 1:
                        const = body[0, 6:9].astype(np.float32).astype(np.float64) if (rnd == 1 and smooth > 1) else body[0, 6:9]
                        assert np.array_equal(msg[s, 21:25], [1.0, 0.0, 0.0, 0.0]) and np.array_equal(msg[s, 18:21], const)
                ref0 = g[f"msg_s{smooth}_mc1"][f]                     # the reference itself, end to end
                worst_ref = max(worst_ref, float(np.abs(msg[0] - ref0[:25]).max()))
                if smooth > 1:
                    assert len(ref0) == 25 + 6 * smooth
                    worst_ref = max(worst_ref, float(np.abs(tail[0].reshape(-1) - ref0[25:]).max()))
            bank.reset()
            wins = [orc.WindowOracle(T, smooth, stats, predict) for _ in range(S)]
        m.check()
        assert worst_orc < 5e-6 and worst_tail < 5e-6 and worst_ref < 5e-6, (smooth, worst_orc, worst_tail, worst_ref)
    with pytest.raises(UserWarning):
        StreamBank(m, S, T, smooth=65)
    with pytest.raises(UserWarning):
        StreamBank(m, S, T).step()                            # nothing pushed yet


@pytest.mark.parametrize("name,S,n_mc,smooth,p_drop", [("pocket", 5, 3, 1, 0.0), ("pocket", 3, 25, 5, 0.2),
                                                        ("watch", 2, 70, 1, 0.2), ("uarm", 4, 4, 3, 0.2)])
def test_stream_bank_monte_carlo(golden, norm_stats, name, S, n_mc, smooth, p_drop):
    """ape_streams_set_mc: n_mc dropout samples per stream and frame inside the bank.  The stacked rows
    (smooth x n_mc per stream, oldest prediction first: estimator.py:112-118), FK, sign-aligned means and tails are
    checked against the oracle fed with the SAME samples, which come from ape_lstm_forward on explicitly repeated
    windows with the bank's Philox key (the dropout arithmetic itself is pinned by the injected-mask tests); with
    p = 0 every sample must equal the deterministic oracle model."""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    g = golden(f"stream_trace_{name}.npz")
    stats, body = norm_stats[name], g["body"]
    m, sd, cfg = make_model(name, int(g["weights_seed"]), stats)
    m.set_body(body)
    T, I, O = cfg["T"], cfg["I"], cfg["O"]
    seed = 0xABCDE12345
    bank = StreamBank(m, S, T, smooth=smooth, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc,
                      dropout=p_drop, seed=seed)
    rng = np.random.default_rng(17)
    F = 9
    feats = _synthetic_windows(stats, S, F, I, 23)                  # [S,F,I] feature rows, frame f of stream s
    lib = _hip.lib()
    flags = _hip.FLAG_NORMALIZE_INPUT | (_hip.FLAG_DROPOUT_PHILOX if p_drop > 0 and cfg["L"] > 1 else 0)
    calls = 0
    worst_msg = worst_tail = 0.0
    for rnd in range(2):
        shadow = [orc.WindowOracle(T, 1, None, lambda h: np.zeros((1, O))) for _ in range(S)]
        samples = {}
        wins = [orc.WindowOracle(T, smooth, None, (lambda h, s=s: samples[s])) for s in range(S)]
        for f in range(F):
            bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda())
            msg, tail = bank.step(with_tail=True)
            msg, tail = msg.cpu().numpy(), tail.cpu().numpy()
            assert msg.shape == (S, 25) and tail.shape == (S, smooth * n_mc, 6)
            # the same windows, repeated n_mc times each, through the plain entry point with the bank's key
            hist = []
            for s in range(S):
                shadow[s].push(feats[s, f])
                hist.append(np.vstack(shadow[s].rows).astype(np.float32))
            x = torch.from_numpy(np.repeat(np.stack(hist), n_mc, axis=0)).cuda()
            y = torch.empty((S * n_mc, O), dtype=torch.float32, device="cuda")
            _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), S * n_mc, T, flags, None, float(p_drop),
                                            seed + calls, C.c_void_p(y.data_ptr()), None), "ape_lstm_forward")
            calls += 1
            torch.cuda.synchronize()
            y = y.cpu().numpy().astype(np.float64) * stats["yy_s"] + stats["yy_m"]
            if p_drop == 0.0:
                det = orc.infer_windows(sd, stats, body, cfg["layout"], np.stack(hist))[0].astype(np.float64) * stats["yy_s"] + stats["yy_m"]
                assert np.abs(y.reshape(S, n_mc, O) - det[:, None, :]).max() < 2e-5
            else:
                assert np.abs(y.reshape(S, n_mc, O) - y.reshape(S, n_mc, O)[:, :1]).max() > 1e-3      # samples differ
            for s in range(S):
                samples[s] = y[s * n_mc:(s + 1) * n_mc]
                pred = wins[s].push(feats[s, f])
                assert pred.shape == (smooth * n_mc, O)
                est = orc.arm_pose_from_targets(pred, body, cfg["layout"], "eigh")
                ref = orc.msg_from_est(est, body, cfg["layout"])
                worst_msg = max(worst_msg, float(np.abs(msg[s] - ref).max()))
                worst_tail = max(worst_tail, float(np.abs(tail[s] - est[:, :6]).max()))
        bank.reset()
    m.check()
    assert worst_msg < 5e-6 and worst_tail < 5e-6, (worst_msg, worst_tail)
    with pytest.raises(UserWarning):
        StreamBank(m, S, T, smooth=64, monte_carlo_samples=65)      # smooth * n_mc > 4096


@pytest.mark.parametrize("name,S,n_mc", [("pocket", 330, 25), ("watch", 330, 25), ("uarm", 170, 50)])
def test_stream_bank_monte_carlo_shared_layer0(golden, norm_stats, name, S, n_mc):
    """From 8192 sample rows on, the bank computes layer 0 ONCE per stream and runs the layers above alone over the S x n_mc
    rows (nn.LSTM's dropout sits between the layers, nn_models.py:169-174, so h_0 is common to a
    stream's samples).  The samples must be the ones a fused launch over the same rows draws: checked against
    ape_lstm_forward on explicitly repeated windows (batch-tile kernel, same Philox key) through the oracle's FK --
    every sample row's hand / elbow position, and the full message of a spread of streams."""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    p_drop, seed = 0.2, 0x1234567
    g = golden(f"stream_trace_{name}.npz")
    stats, body = norm_stats[name], g["body"]
    m, sd, cfg = make_model(name, int(g["weights_seed"]), stats)
    m.set_body(body)
    T, I, O = cfg["T"], cfg["I"], cfg["O"]
    assert S * n_mc >= 8192
    bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=p_drop, seed=seed)
    F = 8                                                           # more frames than the window is long
    feats = _synthetic_windows(stats, S, F, I, 41)
    lib = _hip.lib()
    shadow = [orc.WindowOracle(T, 1, None, lambda h: np.zeros((1, O))) for _ in range(S)]
    worst_tail = worst_msg = 0.0
    for f in range(F):
        bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda())
        msg, tail = bank.step(with_tail=True)
        msg, tail = msg.cpu().numpy(), tail.cpu().numpy()
        hist = []
        for s in range(S):
            shadow[s].push(feats[s, f])
            hist.append(np.vstack(shadow[s].rows).astype(np.float32))
        x = torch.from_numpy(np.repeat(np.stack(hist), n_mc, axis=0)).cuda()
        y = torch.empty((S * n_mc, O), dtype=torch.float32, device="cuda")
        m.set_kernel("tile16")
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), S * n_mc, T,
                                        _hip.FLAG_NORMALIZE_INPUT | _hip.FLAG_DROPOUT_PHILOX, None, p_drop, seed + f,
                                        C.c_void_p(y.data_ptr()), None), "ape_lstm_forward")
        torch.cuda.synchronize()
        m.set_kernel("auto")
        y = y.cpu().numpy().astype(np.float64) * stats["yy_s"] + stats["yy_m"]
        assert np.abs(y.reshape(S, n_mc, O) - y.reshape(S, n_mc, O)[:, :1]).max() > 1e-3          # samples differ
        est = orc.arm_pose_from_targets(y, body, cfg["layout"], "closed")
        worst_tail = max(worst_tail, float(np.abs(tail.reshape(S * n_mc, 6) - est[:, :6]).max()))
        for s in list(range(0, S, 37)) + [S - 1]:
            e = orc.arm_pose_from_targets(y[s * n_mc:(s + 1) * n_mc], body, cfg["layout"], "eigh")
            worst_msg = max(worst_msg, float(np.abs(msg[s] - orc.msg_from_est(e, body, cfg["layout"])).max()))
    m.check()
    assert worst_tail < 5e-6 and worst_msg < 5e-6, (worst_tail, worst_msg)


@pytest.mark.parametrize("T,philox", [(6, False), (6, True), (64, False)])
def test_auto_dispatch_splits_large_batches(norm_stats, T, philox):
    """APE_KERNEL_AUTO on batches of 4096 rows and more: whole 4096-row waves go to the batch-tile kernel
    where that is cheaper (short windows, dropout), the rest to the cluster kernel.  Whatever the split, every row
    must be what ONE of the two kernels yields for it and agree with the oracle."""
    from wear_mocap_ape_amd import _hip
    st = norm_stats["pocket"]
    m, sd, cfg = make_model("pocket", 3, st)
    B = 4096 + 300
    raw = _synthetic_windows(st, B, T, cfg["I"], 31)
    x = torch.from_numpy(raw).cuda()
    lib = _hip.lib()
    flags = _hip.FLAG_NORMALIZE_INPUT | (_hip.FLAG_DROPOUT_PHILOX if philox else 0)
    out = {}
    for kern in ("auto", "tile16", "cluster"):
        m.set_kernel(kern)
        y = torch.zeros((B, cfg["O"]), dtype=torch.float32, device="cuda")
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, flags, None, 0.2 if philox else 0.0, 77,
                                        C.c_void_p(y.data_ptr()), None), "ape_lstm_forward")
        torch.cuda.synchronize()
        m.check()
        out[kern] = y.cpu().numpy()
    m.set_kernel("auto")
    a, t16, cl = out["auto"], out["tile16"], out["cluster"]
    if T == 64 and not philox:
        # round 2: on long eval-mode windows five launches of the second-generation cluster kernel (16 + 12.4 T us each)
        # are priced below a batch-tile wave + one more launch, so AUTO keeps the whole batch on it
        assert np.array_equal(a, cl) and not np.array_equal(a[:4096], t16[:4096])
    else:
        # one batch-tile wave in front, the 300 remaining rows on the cluster kernel
        assert np.array_equal(a[:4096], t16[:4096])
        assert not np.array_equal(a[:4096], cl[:4096])          # (the two kernels do differ in the last bits)
        # the tail ran on the cluster kernel, as a launch of 300 rows: its head sums in the order of that launch's
        # row-tile count, so against the whole-batch cluster run it agrees to rounding, not to the bit
        assert np.abs(a[4096:] - cl[4096:]).max() < 1e-6 and not np.array_equal(a[4096:], t16[4096:])
    # well below a wave's worth of long windows the cluster kernel keeps the whole batch
    n = 2500
    y = torch.zeros((n, cfg["O"]), dtype=torch.float32, device="cuda")
    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), n, T, flags, None, 0.2 if philox else 0.0, 77,
                                    C.c_void_p(y.data_ptr()), None), "ape_lstm_forward")
    torch.cuda.synchronize()
    if T == 64:                         # (2500 rows of SHORT windows cost about one batch-tile wave either way)
        assert np.abs(y.cpu().numpy() - cl[:n]).max() < 1e-6 and not np.array_equal(y.cpu().numpy(), t16[:n])
    if not philox:
        pick = np.r_[0:8, 4090:4104, B - 8:B]
        ref = orc.infer_windows(sd, st, orc.DEFAULT_BODY, cfg["layout"], raw[pick])[0]
        assert np.abs(a[pick] - ref).max() < (TOL_Y_T64 if T == 64 else 1e-5)
    else:               # the two kernels draw the same masks for the same (row, step, unit) within a launch
        assert np.abs(t16[:512] - cl[:512]).max() < 1e-5


@pytest.mark.parametrize("name,B", [("pocket", 2500), ("uarm", 4100)])
def test_batches_beyond_one_cluster_launch(norm_stats, name, B):
    """more windows than one cluster launch covers (1024 / 2048 rows): the entry point chunks the batch; every chunk
    must read its own inputs and write its own outputs (spot-checked against the oracle, all rows against tile16)"""
    st = norm_stats[name]
    m, sd, cfg = make_model(name, 11, st)
    raw = _synthetic_windows(st, B, cfg["T"], cfg["I"], 9)
    x = torch.from_numpy(raw).cuda()
    y = m(x, last_step_only=True, normalize_input=True)[:, 0].cpu().numpy()
    m.check()
    y16 = m.set_kernel("tile16")(x, last_step_only=True, normalize_input=True)[:, 0].cpu().numpy()
    m.set_kernel("auto")
    assert np.abs(y - y16).max() < TOL_Y_SHORT
    pick = np.r_[0:3, 1022:1027, 2046:2051, B - 3:B]
    xn = ((raw[pick].astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
    assert np.abs(y[pick] - orc.lstm_forward(sd, xn)[:, -1]).max() < TOL_Y_SHORT



def test_c_caller(tmp_path):
    """tests/c_abi/demo.c -- plain C on the C ABI, no Python or torch in the process -- against the oracle on the same
    LCG-seeded weights, statistics and windows: ape_infer rows and the stream bank's messages"""
    import subprocess
    from tests.test_host_bookkeeping import build_c_caller
    exe = build_c_caller(tmp_path)
    out = tmp_path / "demo.bin"
    run = subprocess.run([str(exe), str(out)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr + run.stdout
    I, H, L, O, B, T, S, SMOOTH = 22, 256, 2, 14, 37, 6, 5, 3

    def lcg(seed, n):                              # the generator of demo.c
        state, vals = np.uint32(seed), np.empty(n, np.float32)
        s = int(state)
        for i in range(n):
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            vals[i] = np.float32(s >> 8) * np.float32(2.0 / 16777216.0) - np.float32(1.0)
        return vals

    from wear_mocap_ape_amd.estimate import nn_models
    n_w = 4 * H * (I + H) + 8 * H + 4 * H * (H + H) + 8 * H + O * H + O
    blob = (np.float32(0.0625) * lcg(12345, n_w)).astype(np.float32)
    sd, cur = {}, 0
    for key in nn_models.state_dict_keys(L):
        shape = {"weight_ih_l0": (4 * H, I), "weight_ih_l1": (4 * H, H), "weight_hh_l0": (4 * H, H), "weight_hh_l1": (4 * H, H),
                 "bias_ih_l0": (4 * H,), "bias_hh_l0": (4 * H,), "bias_ih_l1": (4 * H,), "bias_hh_l1": (4 * H,),
                 "weight": (O, H), "bias": (O,)}[key.split(".")[-1]]
        n = int(np.prod(shape))
        sd[key] = blob[cur:cur + n].reshape(shape)
        cur += n
    assert cur == n_w
    stats = {"xx_m": 0.1 * np.arange(I) - 1.0, "xx_s": 0.5 + 0.05 * np.arange(I),
             "yy_m": 0.02 * np.arange(O), "yy_s": 0.3 + 0.01 * np.arange(O)}
    u = lcg(777, B * T * I).reshape(B, T, I)
    x = (stats["xx_m"] + stats["xx_s"] * u.astype(np.float64)).astype(np.float32)      # (float)(m + s * u) in double, as in C
    raw = np.fromfile(out, dtype=np.uint8)
    y = raw[:B * O * 4].view(np.float32).reshape(B, O)
    est = raw[B * O * 4:B * O * 4 + B * 21 * 8].view(np.float64).reshape(B, 21)
    o_msg = B * O * 4 + B * 21 * 8
    msg = raw[o_msg:o_msg + S * 25 * 8].view(np.float64).reshape(S, 25)
    F, DG = 9, 25 + 6 * SMOOTH
    feats = raw[o_msg + S * 25 * 8:o_msg + S * 25 * 8 + F * I * 4].view(np.float32).reshape(F, I)
    dgram = raw[o_msg + S * 25 * 8 + F * I * 4:].view(np.float32).reshape(F, DG)
    y_ref, est_ref = orc.infer_windows(sd, stats, orc.DEFAULT_BODY, 0, x, route="eigh")
    assert np.abs(y - y_ref).max() < TOL_Y_SHORT
    assert np.abs(est - est_ref).max() < TOL_EST_E2E
    # stream bank: T frames pushed, smoothing stack of the last 3 predictions; frame f sees the window padded with x_0
    predict = lambda hist: orc.lstm_forward(sd, np.asarray(hist, dtype=np.float32)[None])[:, -1, :]
    for s in range(S):
        win = orc.WindowOracle(T, SMOOTH, stats, predict)
        for t in range(T):
            pred = win.push(x[s, t])
        ref = orc.msg_from_est(orc.arm_pose_from_targets(pred, orc.DEFAULT_BODY, 0, "eigh"), orc.DEFAULT_BODY, 0)
        assert np.abs(msg[s] - ref).max() < 5e-6
    # ape_streams_frame_host (ABI 6): one estimator, F frames, host rows in / host datagrams out; the rows' features (ape_parse_rows,
    # pinned to the reference by the parse goldens) through the oracle's window, model, smoothing stack and message + tail
    assert np.isfinite(feats).all() and np.isfinite(dgram).all()
    win = orc.WindowOracle(T, SMOOTH, stats, predict)
    for fr in range(F):
        pred = win.push(feats[fr])
        est = orc.arm_pose_from_targets(pred, orc.DEFAULT_BODY, 0, "eigh")
        ref = np.asarray(orc.msg_with_mc_samples(orc.msg_from_est(est, orc.DEFAULT_BODY, 0), est, True), dtype=np.float64)
        assert ref.shape == (DG,) and np.abs(dgram[fr] - ref).max() < 1e-5, (fr, float(np.abs(dgram[fr] - ref).max()))


def test_two_models_on_two_streams_concurrently(norm_stats):
    """two cluster-kernel launches (different models, own exchange buffers and tickets) in flight on the same GPU at
    once: their workgroups compete for the CUs, clusters form by arrival ticket, nothing deadlocks, results are the
    ones each model produces alone"""
    a, _, cfg_a = make_model("pocket", 31, norm_stats["pocket"])
    b, _, cfg_b = make_model("uarm", 32, norm_stats["uarm"])
    xa = torch.from_numpy(_synthetic_windows(norm_stats["pocket"], 1024, 16, cfg_a["I"], 1)).cuda()
    xb = torch.from_numpy(_synthetic_windows(norm_stats["uarm"], 2048, 16, cfg_b["I"], 2)).cuda()
    ya_alone = a(xa, last_step_only=True, normalize_input=True).clone()
    yb_alone = b(xb, last_step_only=True, normalize_input=True).clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(25):
        with torch.cuda.stream(sa):
            ya = a(xa, last_step_only=True, normalize_input=True)
        with torch.cuda.stream(sb):
            yb = b(xb, last_step_only=True, normalize_input=True)
    torch.cuda.synchronize()
    a.check()
    b.check()
    assert torch.equal(ya, ya_alone) and torch.equal(yb, yb_alone)


@pytest.mark.parametrize("name,B,T", [("pocket", 1024, 64), ("pocket", 3, 2), ("uarm", 300, 6), ("watch", 70, 8)])
def test_all_steps_output_on_the_cluster_kernel(norm_stats, name, B, T):
    """DropoutLSTM.forward returns every step (nn_models.py:188-189): under AUTO that is the cluster kernel writing each
    step's top-layer output to [B,T,H] plus one head launch over those rows -- against the batch-tile kernel (every
    step) and the oracle (sampled windows), with the z-score fused; the last step equals the last-step-only call"""
    st = norm_stats[name]
    m, sd, cfg = make_model(name, 41, st)
    raw = _synthetic_windows(st, B, T, cfg["I"], 17)
    x = torch.from_numpy(raw).cuda()
    y_all = m(x, normalize_input=True)
    assert tuple(y_all.shape) == (B, T, cfg["O"])
    m.check()
    y_last = m(x, last_step_only=True, normalize_input=True)
    assert np.abs(y_all[:, -1].cpu().numpy() - y_last[:, 0].cpu().numpy()).max() < TOL_Y_SHORT
    y16 = m.set_kernel("tile16")(x, normalize_input=True)
    m.set_kernel("auto")
    assert np.abs(y_all.cpu().numpy() - y16.cpu().numpy()).max() < TOL_Y_SHORT
    pick = np.unique(np.r_[0:2, B // 2, B - 2:B])
    xn = ((raw[pick].astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
    assert np.abs(y_all.cpu().numpy()[pick] - orc.lstm_forward(sd, xn)).max() < TOL_Y_SHORT
