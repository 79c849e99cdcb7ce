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


@pytest.mark.parametrize("name,S,dtype", [("pocket", 1024, torch.float32), ("pocket", 600, torch.float64), ("watch", 513, torch.float32)])
def test_the_eval_banks_post_filter_in_the_regressors_tail(norm_stats, name, S, dtype):
    """Round 6: a deterministic bank without stacking (smooth = 1, one row per stream) whose step runs on `ape_lstm_cluster32`'s short-window
    instantiation gets its post-filter (de-normalise, FK, message: estimator.py:108-137) in the TAIL of that launch -- a member finishes four
    windows in the head and its four waves are the post-filter's four role waves for them (`stream_post_lanes<TMsg, 4>`): two kernels per
    frame instead of three.  Held to the oracle on sampled streams (window bookkeeping -> LSTM -> FK -> message, 5e-6) and, on EVERY stream,
    bit for bit to the three-kernel frame (a profiled bank keeps the kernels apart; the same device functions in the same order); ragged
    stream counts (the last cluster's members own streams past the bank), both message types, the tail, the packed host frame (whose status
    word the last workgroup out writes) included."""
    import ctypes as C
    from tests.test_hip_parity import make_model
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    st = norm_stats[name]
    m, sd, cfg = make_model(name, 21, st)
    body = orc.DEFAULT_BODY
    m.set_body(body)
    T, I, O = cfg["T"], cfg["I"], cfg["O"]
    rng = np.random.default_rng(S)
    fused, apart = StreamBank(m, S, T, smooth=1, normalize=True, dtype=dtype), StreamBank(m, S, T, smooth=1, normalize=True, dtype=dtype)
    apart.profile(True)
    ring = None
    sample = sorted(set([0, 1, 3, 4, 31, 32, S - 5, S - 4, S - 1] + rng.integers(0, S, 24).tolist()))
    worst = worst_tail = 0.0
    for f in range(T + 3):
        xx = (st["xx_m"] + st["xx_s"] * rng.normal(size=(S, I))).astype(np.float32)
        ring = np.repeat(xx[:, None, :], T, axis=1) if ring is None else np.concatenate([ring[:, 1:], xx[:, None, :]], axis=1)
        xd = torch.from_numpy(xx).cuda()
        fused.push_features(xd); apart.push_features(xd)
        mf, tf = fused.step(with_tail=True)
        assert m.last_kernel() == "ape_lstm_cluster32", m.last_kernel()
        ma, ta = apart.step(with_tail=True)
        mf, tf, ma, ta = mf.cpu().numpy(), tf.cpu().numpy(), ma.cpu().numpy(), ta.cpu().numpy()
        assert mf.dtype == (np.float32 if dtype == torch.float32 else np.float64)
        assert np.array_equal(mf, ma) and np.array_equal(tf, ta), f
        xn = ((ring.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
        pred = orc.lstm_forward(sd, xn[sample])[:, -1].astype(np.float64) * st["yy_s"] + st["yy_m"]
        for k, s in enumerate(sample):
            est = orc.arm_pose_from_targets(pred[k][None], body, cfg["layout"], "eigh")
            worst = max(worst, float(np.abs(mf[s] - orc.msg_from_est(est, body, cfg["layout"])).max()))
            worst_tail = max(worst_tail, float(np.abs(tf[s] - est[:, :6]).max()))
    m.check()
    assert worst < 5e-6 and worst_tail < 5e-6, (worst, worst_tail)
    assert apart.profile_read()[1] == T + 3            # the profiled bank did run its regressor as a launch of its own
    # the host frame (packed rows: message + tail; raw messages in): the same two banks, one more frame
    kind = {"pocket": _hip.PARSE_WATCH_PHONE_POCKET, "watch": _hip.PARSE_WATCH_ONLY}[name]
    rows = rng.normal(size=(S, _hip.PARSE_SHAPES[kind][0])).astype(np.float32)
    outs = []
    for bank in (fused, apart):
        out = np.zeros((S, 31), np.float64 if dtype == torch.float64 else np.float32)
        _hip.check(_hip.lib().ape_streams_frame_host(bank._handle, kind, rows.ctypes.data_as(C.c_void_p), _hip.FLAG_NORMALIZE_INPUT,
                                                     out.ctypes.data_as(C.c_void_p), _hip.F64 if dtype == torch.float64 else _hip.F32, None),
                   "ape_streams_frame_host")
        outs.append(out)
    assert np.array_equal(outs[0], outs[1]) and np.isfinite(outs[0]).all() and np.abs(outs[0][:, :4]).max() > 0.0
    m.check()
