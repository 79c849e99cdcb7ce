"""GPU parity tests added in round 2 (all through the C ABI of libape_hip.so):
  * BASELINE configs[3]: 8192 streams sharded over G ranks == the unsharded call, bit for bit,
  * Monte-Carlo dropout against the DISTRIBUTION of the reference's own `monte_carlo_predictions` samples
    (tests/golden/mc_stats.npz, drawn by the reference, nn_models.py:191-207),
  * `DropoutLSTM.forward(x, hs=(h0, c0))` against the reference's outputs (tests/golden/lstm_hs.npz),
  * an aborted cluster launch fails loudly and the handle recovers.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ape_oracle as orc
from tests import mc_check
from tests.test_hip_parity import make_model, _synthetic_windows, quat_err, TOL_Y_SHORT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as entry
    entry.build()


# ---------------- configs[3]: the 8192-stream split -------------------------------------------------------------
@pytest.mark.parametrize("T", [6, 64])
def test_sharded_streams_equal_unsharded(norm_stats, T):
    """8192 independent streams on G ranks (contiguous `shard_range`s, one GPU each) must give what ONE call over all
    8192 windows gives: windows are independent (zero state per window, nn_models.py:180-189), so the concatenation of
    the shards' outputs is bit-equal to the unsharded output when every shard runs the kernel the unsharded call
    ran for those rows.  G = 2 (4096 rows per rank: one batch-tile wave each, like the unsharded call's two) and
    G = 8 (1024 rows per rank: the cluster kernel; the unsharded call is pinned to the same kernel), plus the oracle
    on a sampled subset.  The shards run one after the other on this one GPU, each on its own model handle that got
    the weights the way a rank gets them (flat blob, `load_weight_blob`)."""
    from wear_mocap_ape_amd import _hip, streams
    from wear_mocap_ape_amd.estimate import nn_models
    name, S = "pocket", 8192
    stats = norm_stats[name]
    m_all, sd, cfg = make_model(name, 0, stats)
    m_all.set_body(orc.DEFAULT_BODY)
    x = _synthetic_windows(stats, S, T, cfg["I"], 5)
    xd = torch.from_numpy(x).cuda()
    lib = _hip.lib()

    def run(model, xs):
        n = xs.shape[0]
        y = torch.empty((n, cfg["O"]), dtype=torch.float32, device="cuda")
        est = torch.empty((n, 21), dtype=torch.float64, device="cuda")
        _hip.check(lib.ape_infer(model.handle, C.c_void_p(xs.data_ptr()), n, T, _hip.FLAG_NORMALIZE_INPUT,
                                 C.c_void_p(y.data_ptr()), C.c_void_p(est.data_ptr()), _hip.F64, None), "ape_infer")
        torch.cuda.synchronize()
        model.check()
        return y.cpu().numpy(), est.cpu().numpy()

    blob = streams.flatten_state_dict(sd, nn_models.state_dict_keys(cfg["L"]))
    for G, kernel in ((2, "auto"), (8, "cluster"), (8, "tile16")):
        m_all.set_kernel(kernel)
        y_all, est_all = run(m_all, xd)
        ys, es = [], []
        for r in range(G):
            lo, hi = streams.shard_range(S, r, G)
            assert hi - lo == S // G
            m_r = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=0)
            m_r.load_weight_blob(torch.from_numpy(blob).cuda())           # what the RCCL broadcast leaves on a rank
            m_r.set_norm_stats(stats["xx_m"], stats["xx_s"], stats["yy_m"], stats["yy_s"])
            m_r.set_body(orc.DEFAULT_BODY)
            m_r.set_kernel(kernel)
            y_r, e_r = run(m_r, xd[lo:hi].contiguous())
            ys.append(y_r)
            es.append(e_r)
        y_cat, e_cat = np.concatenate(ys), np.concatenate(es)
        assert np.array_equal(y_cat, y_all), (G, kernel, float(np.abs(y_cat - y_all).max()))
        assert np.array_equal(e_cat, est_all), (G, kernel)
        # and the oracle on a sample of the streams (tolerances of the module header)
        idx = np.r_[0:3, 1023:1026, 4095:4098, 8189:8192]
        y_ref, est_ref = orc.infer_windows(sd, stats, orc.DEFAULT_BODY, cfg["layout"], x[idx])
        assert np.abs(y_all[idx] - y_ref).max() < 1e-6
        assert np.abs(est_all[idx][:, :9] - est_ref[:, :9]).max() < 2e-6
        for c in (9, 13, 17):
            assert quat_err(est_all[idx][:, c:c + 4], est_ref[:, c:c + 4]) < 2e-6
    m_all.set_kernel("auto")


def test_sharded_stream_banks_equal_one_bank(norm_stats):
    """the same for the device-side stream bank (window rings, smoothing, messages): 8192 streams in one bank vs
    8 banks of 1024, four frames from a cold start; eval mode, so no random stream is involved"""
    from wear_mocap_ape_amd import _hip, streams
    from wear_mocap_ape_amd.streams import StreamBank
    name, S, G, T = "pocket", 8192, 8, 6
    stats = norm_stats[name]
    m, sd, cfg = make_model(name, 0, stats)
    m.set_body(orc.DEFAULT_BODY)
    m.set_kernel("cluster")                     # 8192 rows would otherwise go to the batch-tile kernel, 1024 to this one
    rng = np.random.default_rng(9)
    rows = rng.normal(size=(4, S, 55)).astype(np.float32)
    whole = StreamBank(m, S, T, smooth=3, normalize=True, dtype=torch.float32)
    parts = [StreamBank(m, S // G, T, smooth=3, normalize=True, dtype=torch.float32) for _ in range(G)]
    for f in range(4):
        whole.push_rows(torch.from_numpy(rows[f]).cuda(), _hip.PARSE_WATCH_PHONE_POCKET)
        msg_all = whole.step().cpu().numpy().copy()
        got = []
        for r in range(G):
            lo, hi = streams.shard_range(S, r, G)
            parts[r].push_rows(torch.from_numpy(rows[f, lo:hi]).cuda(), _hip.PARSE_WATCH_PHONE_POCKET)
            got.append(parts[r].step().cpu().numpy().copy())
        assert np.array_equal(np.concatenate(got), msg_all), f
    m.check()
    m.set_kernel("auto")


# ---------------- Monte-Carlo dropout vs the reference's own sample distribution ----------------------------------
@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_monte_carlo_predictions_match_reference_distribution(golden, name):
    """`model.monte_carlo_predictions(n, x)` (in-kernel Philox masks) against mean / variance / quantiles /
    correlations of 24 000 samples the REFERENCE drew for the same weights and windows (nn_models.py:191-207:
    self.lstm.train() + x.repeat).  n = 24 000 rows under APE_KERNEL_AUTO (whole batch-tile waves for a sample this
    large), then each kernel on its own with 8192 rows (the cluster kernel in 16 launches of 512 rows, each with its
    own Philox key)."""
    g = golden("mc_stats.npz")
    cfg = orc.MODEL_CONFIGS[name]
    m, sd, _ = make_model(name, 0)
    p, n_ref, levels = float(g[f"dropout_{name}"]), int(g["n_samples"]), g["quantile_levels"]
    assert abs(m.dropout - p) < 1e-12
    m.manual_seed(2024)
    for w in range(3):
        x = torch.from_numpy(g[f"x_{name}"][w:w + 1]).cuda()
        args = (g[f"y_mean_{name}"][w], g[f"y_cov_{name}"][w], g[f"y_quant_{name}"][w], levels, n_ref)
        y = m.monte_carlo_predictions(24000, x, last_step_only=True)
        assert tuple(y.shape) == (24000, 1, cfg["O"]) and m.lstm.training
        bad = mc_check.compare(y[:, 0].cpu().numpy(), *args, what=f"{name} w{w} auto")
        assert not bad, bad
        for kernel, n in (("tile16", 8192), ("cluster", 8192)):
            m.set_kernel(kernel)
            yk = m.monte_carlo_predictions(n, x, last_step_only=True)[:, 0].cpu().numpy()
            m.set_kernel("auto")
            bad = mc_check.compare(yk, *args, what=f"{name} w{w} {kernel}")
            assert not bad, bad
    # negative control on the device path: the same sampler with a different rate must be flagged
    m.dropout = 0.5 * p
    y = m.monte_carlo_predictions(24000, torch.from_numpy(g[f"x_{name}"][0:1]).cuda(), last_step_only=True)
    assert mc_check.compare(y[:, 0].cpu().numpy(), g[f"y_mean_{name}"][0], g[f"y_cov_{name}"][0], g[f"y_quant_{name}"][0],
                            levels, n_ref)
    m.check()


@pytest.mark.parametrize("name,S,n_mc", [("pocket", 375, 64), ("uarm", 375, 64)])
def test_stream_bank_mc_matches_reference_distribution(golden, name, S, n_mc):
    """the stream bank's Monte-Carlo mode on its shared-layer-0 route (layer 0 once per stream, layers above over the
    S x n_mc sample rows): S streams are fed the SAME window, so their S * n_mc = 24 000 hand / elbow
    positions (message tail) are samples of one window's distribution -- compared with the statistics of the
    reference's samples pushed through the reference's FK (mc_stats.npz `est6_*`)."""
    from wear_mocap_ape_amd.streams import StreamBank
    g = golden("mc_stats.npz")
    cfg = orc.MODEL_CONFIGS[name]
    m, sd, _ = make_model(name, 0)
    m.set_body(orc.DEFAULT_BODY)
    p, n_ref, levels = float(g[f"dropout_{name}"]), int(g["n_samples"]), g["quantile_levels"]
    assert S * n_mc >= 8192
    for w in (0, 2):
        x = g[f"x_{name}"][w]                                       # [T, I] normalised model input
        bank = StreamBank(m, S, cfg["T"], smooth=1, normalize=False, dtype=torch.float64, monte_carlo_samples=n_mc,
                          dropout=p, seed=99 + w)
        for t in range(cfg["T"]):
            bank.push_features(torch.from_numpy(np.repeat(x[t][None], S, axis=0)).cuda())
        msg, tail = bank.step(with_tail=True)
        tail = tail.cpu().numpy().reshape(S * n_mc, 6)
        bad = mc_check.compare(tail, g[f"est6_mean_{name}"][w], g[f"est6_cov_{name}"][w], g[f"est6_quant_{name}"][w], levels,
                               n_ref, what=f"{name} w{w} bank")
        assert not bad, bad
        del bank
    m.check()


def test_mlp_monte_carlo_matches_reference_distribution(golden):
    """DropoutFF.monte_carlo_predictions (dropout in front of the output layer, nn_models.py:356-370)"""
    from wear_mocap_ape_amd.estimate import nn_models
    g = golden("mc_stats.npz")
    I, H, n_hidden, O = (int(v) for v in g["dims_ff"])
    sd = orc.make_ff_state_dict(I, H, n_hidden, O, 0)
    m = nn_models.DropoutFF(output_size=O, hidden_layer_size=H, hidden_layer_count=n_hidden, input_size=I, dropout=0.2, device=0)
    m.load_state_dict(sd)
    m.manual_seed(7)
    y = m.monte_carlo_predictions(int(g["n_samples"]), torch.from_numpy(g["x_ff"]).cuda(), last_step_only=True)
    bad = mc_check.compare(y[:, 0].cpu().numpy(), g["y_mean_ff"][0], g["y_cov_ff"][0], g["y_quant_ff"][0],
                           g["quantile_levels"], int(g["n_samples"]), what="ff")
    assert not bad, bad


# ---------------- non-zero initial state ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["pocket", "watch", "uarm"])
def test_initial_state_vs_reference_golden(golden, name):
    """DropoutLSTM.forward(x, hs=(h0, c0)) (nn_models.py:180-189) against the reference module's outputs"""
    g = golden("lstm_hs.npz")
    m, sd, cfg = make_model(name, 0)
    for (B, T) in ((1, cfg["T"]), (5, cfg["T"]), (3, 64), (18, 2)):
        k = f"{name}_B{B}_T{T}"
        hs = (torch.from_numpy(g["h0_" + k]), torch.from_numpy(g["c0_" + k]))
        y = m(torch.from_numpy(g["x_" + k]), hs)                       # host in -> host out, all steps
        assert tuple(y.shape) == (B, T, cfg["O"])
        assert np.abs(y.numpy() - g["y_" + k]).max() < 1e-6, (name, B, T)
        y_last = m(torch.from_numpy(g["x_" + k]).cuda(), (hs[0].cuda(), hs[1].cuda()), last_step_only=True)
        assert np.abs(y_last.cpu().numpy()[:, 0] - g["y_" + k][:, -1]).max() < 1e-6
        # zero state given explicitly == no state given (any kernel)
        z = torch.zeros_like(hs[0])
        assert np.abs(m(torch.from_numpy(g["x_" + k]), (z, z)).numpy() - m(torch.from_numpy(g["x_" + k])).numpy()).max() < 1e-6
    with pytest.raises(UserWarning):
        m(torch.from_numpy(g[f"x_{name}_B5_T{cfg['T']}"]), (torch.zeros(1, 5, cfg["H"]), torch.zeros(1, 5, cfg["H"])))
    with pytest.raises(UserWarning):
        m(torch.from_numpy(g[f"x_{name}_B5_T{cfg['T']}"]), torch.zeros(cfg["L"], 5, cfg["H"]))


# ---------------- aborted launches are loud and recoverable ----------------------------------------------------------
# ---------------- fp16 kernel, second generation (row-set pipelined, 8-member clusters) ----------------------------
@pytest.mark.parametrize("name,B,T", [("watch", 1024, 64), ("pocket", 700, 8), ("watch", 257, 3), ("pocket", 2081, 6),
                                      ("watch", 1024, 1)])
def test_fp16_second_generation_kernel(norm_stats, name, B, T):
    """configs[4] on lstm_cluster_f16v2.hip (batches above 256 rows): against the oracle's binary16-storage emulation
    (layout / indexing), the float32 oracle (stated tolerance 5e-3), the first-generation fp16 kernel (same arithmetic:
    differences only from float32 summation order), ragged and multi-launch batches, the forced any-placement
    (write-through) exchange (same bits as the in-L2 form), and run-to-run determinism (self-cleaning state)."""
    from wear_mocap_ape_amd import _hip
    st = norm_stats[name]
    model, sd, cfg = make_model(name, 0, st)
    x = _synthetic_windows(st, B, T, cfg["I"], 8)
    xd = torch.from_numpy(x).cuda()
    xn = ((x.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
    model.set_precision("f16")
    assert model.kernel_name(B, T) == "ape_lstm_cluster_f16v2"
    y2 = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    model.check()
    y2b = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    assert np.array_equal(y2, y2b)
    y_wt = torch.empty((B, cfg["O"]), dtype=torch.float32, device="cuda")
    # (the default hand-over is write-through since round 5; the opt-in plain in-XCD form must give the same bits)
    _hip.check(_hip.lib().ape_lstm_forward(model.handle, C.c_void_p(xd.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT | _hip.FLAG_IN_XCD_PLAIN,
                                           None, 0.0, 0, C.c_void_p(y_wt.data_ptr()), None), "ape_lstm_forward")
    torch.cuda.synchronize()
    model.check()
    assert np.array_equal(y_wt.cpu().numpy(), y2)
    # round 4: the 16-unit-member form of the same kernel (APE_FLAG_ALT_FORM: 16-member clusters, two workgroups per CU) -- measured
    # slower and not the default, kept selectable; same arithmetic up to the summation order inside the exchange's k-groups
    y_duo = torch.empty((B, cfg["O"]), dtype=torch.float32, device="cuda")
    _hip.check(_hip.lib().ape_lstm_forward(model.handle, C.c_void_p(xd.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT | _hip.FLAG_ALT_FORM,
                                           None, 0.0, 0, C.c_void_p(y_duo.data_ptr()), None), "ape_lstm_forward")
    torch.cuda.synchronize()
    model.check()
    assert model.last_kernel() == "ape_lstm_cluster_f16v2<duo>"
    assert float(np.abs(y_duo.cpu().numpy() - y2).max()) < 3e-5
    model.set_precision("f16_gen1")
    assert model.kernel_name(B, T) == "ape_lstm_cluster_f16"
    y1 = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    model.set_precision("f32")
    y_ref = orc.lstm_forward(sd, xn)[:, -1]
    y_emu = orc.lstm_forward(sd, xn, storage="f16")[:, -1]
    e_ref, e_emu, e_gen = (float(np.abs(y2 - v).max()) for v in (y_ref, y_emu, y1))
    print(f"\n[{name} B={B} T={T} fp16 v2] vs f32 oracle {e_ref:.2e}, vs f16-emulating oracle {e_emu:.2e}, vs gen-1 kernel {e_gen:.2e}")
    assert e_ref < 5e-3 and e_emu < 3e-4 and e_gen < 3e-4


# ---------------- f32 cluster kernel, second generation (32x32x2 MFMA chain, 8-member clusters) ----------------------
@pytest.mark.parametrize("name,B,T", [("pocket", 1024, 64), ("watch", 700, 8), ("pocket", 513, 3), ("watch", 2081, 6),
                                      ("pocket", 1024, 1), ("pocket", 1024, 2), ("pocket", 600, 4), ("watch", 513, 5),
                                      ("pocket", 1024, 12), ("watch", 1024, 13)])
def test_f32_second_generation_cluster_kernel(norm_stats, name, B, T):
    """lstm_cluster32.hip (eval-mode batches above 512 rows of the 2 x 256 models) against the float32 oracle (module
    tolerance 1e-6), the first-generation cluster kernel (other summation order only), ragged and multi-launch batches,
    the opt-in plain in-XCD exchange (same bits as the default write-through form) and run-to-run determinism.  Windows of up to 8
    steps run the instantiation with the end forms of round 4 (step 0 in front of the weights, MODE 1 / MODE 2 sections), T = 8 | 12 sits on either side of
    the boundary."""
    from wear_mocap_ape_amd import _hip
    st = norm_stats[name]
    model, sd, cfg = make_model(name, 3, st)
    x = _synthetic_windows(st, B, T, cfg["I"], 11)
    xd = torch.from_numpy(x).cuda()
    xn = ((x.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
    model.set_kernel("cluster")
    assert model.kernel_name(B, T) == f"ape_lstm_cluster32<256, 2, 32, {'true' if T <= 8 else 'false'}>"
    y2 = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    model.check()
    y2b = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    assert np.array_equal(y2, y2b)
    y_wt = torch.empty((B, cfg["O"]), dtype=torch.float32, device="cuda")
    # (the default hand-over is write-through since round 5; the opt-in plain in-XCD form must give the same bits)
    _hip.check(_hip.lib().ape_lstm_forward(model.handle, C.c_void_p(xd.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT | _hip.FLAG_IN_XCD_PLAIN,
                                           None, 0.0, 0, C.c_void_p(y_wt.data_ptr()), None), "ape_lstm_forward")
    torch.cuda.synchronize()
    model.check()
    assert np.array_equal(y_wt.cpu().numpy(), y2)
    model.set_kernel("cluster_gen1")
    assert "ape_lstm_cluster<" in model.kernel_name(B, T)
    y1 = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    model.set_kernel("auto")
    y_ref = orc.lstm_forward(sd, xn)[:, -1]
    e_ref, e_gen = float(np.abs(y2 - y_ref).max()), float(np.abs(y2 - y1).max())
    print(f"\n[{name} B={B} T={T} cluster32] vs oracle {e_ref:.2e}, vs gen-1 kernel {e_gen:.2e}")
    assert e_ref < 1e-6 and e_gen < 1e-6


# ---------------- MLP regressor, 32-row workgroups (batches that fill the chip) ------------------------------------------
@pytest.mark.parametrize("n_hidden,N", [(2, 16384 + 37), (1, 20000)])
def test_mlp_regressor_large_batches(n_hidden, N):
    """DropoutFF on the 32-row-per-workgroup instantiation of ape_mlp_tile16 (N >= 64 rows per CU): against the oracle's
    restatement of nn_models.py:340-354 incl. an injected dropout mask in front of the output layer, ragged last workgroup;
    the 16-row instantiation (same model, small slice of the rows) must give the same bits for its rows"""
    from wear_mocap_ape_amd.estimate import nn_models
    I, H, O = 22, 256, 14
    sd = orc.make_ff_state_dict(I, H, n_hidden, O, 5)
    m = nn_models.DropoutFF(output_size=O, hidden_layer_size=H, hidden_layer_count=n_hidden, input_size=I, dropout=0.2, device=0)
    m.load_state_dict(sd)
    m.set_kernel("tile16")                                  # (AUTO hands eval batches of this size to the pipeline kernel, below)
    rng = np.random.default_rng(17)
    x = rng.normal(size=(N, I)).astype(np.float32)
    mask = ((rng.random((N, H)) >= 0.2) / 0.8).astype(np.float32)
    y = m(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(y - orc.ff_forward(sd, x)).max() < 2e-6
    ym = m(torch.from_numpy(x).cuda(), masks=torch.from_numpy(mask).cuda()).cpu().numpy()
    assert np.abs(ym - orc.ff_forward(sd, x, mask=mask)).max() < 2e-6
    y_small = m(torch.from_numpy(x[:100]).cuda()).cpu().numpy()              # 16-row workgroups
    assert np.array_equal(y_small, y[:100])


# ---------------- MLP regressor, the two-stage weight-stationary pipeline (eval mode, chip-filling batches) ---------------
@pytest.mark.gpu
def test_mlp_pipeline_kernel():
    """ape_mlp_pipe (AUTO, eval mode, N >= 64 rows per CU) against the tile kernel (same arithmetic up to the float32 summation
    order) and the oracle's restatement of nn_models.py:340-354: a ragged last tile, fewer tiles than some pairs' share, many
    tiles per pair (the ring wraps), the fused float64 z-score, a [B,T,I] input read at its last step, back-to-back launches
    (the kernel leaves its own hand-over words zeroed), and dropout falling back to the tile kernel"""
    from wear_mocap_ape_amd.estimate import nn_models
    I, H, O = 22, 256, 14
    sd = orc.make_ff_state_dict(I, H, 2, O, 9)
    m = nn_models.DropoutFF(output_size=O, hidden_layer_size=H, hidden_layer_count=2, input_size=I, dropout=0.2, device=0)
    m.load_state_dict(sd)
    rng = np.random.default_rng(23)
    st = {"xx_m": 0.1 * np.arange(I) - 1.0, "xx_s": 0.5 + 0.05 * np.arange(I), "yy_m": np.zeros(O), "yy_s": np.ones(O)}
    m.set_norm_stats(st["xx_m"], st["xx_s"], st["yy_m"], st["yy_s"])
    for N in (16384, 16384 + 37, 40000, 262144 + 5):
        x = rng.normal(size=(N, I)).astype(np.float32)
        xt = torch.from_numpy(x).cuda()
        y_tile = m.set_kernel("tile16")(xt).cpu().numpy()
        y_pipe = m.set_kernel("auto")(xt).cpu().numpy()
        y_again = m(xt).cpu().numpy()
        m.check()
        assert np.abs(y_pipe - y_tile).max() < 1e-6
        assert np.array_equal(y_pipe, y_again)
        if N <= 40000:
            assert np.abs(y_pipe - orc.ff_forward(sd, x)).max() < 2e-6
    # fused z-score (float64, like the tile kernel) and a [B,T,I] input at its last step
    B, T = 20000, 3
    raw = (rng.normal(size=(B, T, I)) * st["xx_s"] + st["xx_m"]).astype(np.float32)
    rt = torch.from_numpy(raw).cuda()
    y_tile = m.set_kernel("tile16")(rt, last_step_only=True, normalize_input=True).cpu().numpy()
    y_pipe = m.set_kernel("auto")(rt, last_step_only=True, normalize_input=True).cpu().numpy()
    m.check()
    assert y_pipe.shape == (B, 1, O) and np.abs(y_pipe - y_tile).max() < 1e-6
    z = ((raw[:, -1].astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
    assert np.abs(y_pipe[:, 0] - orc.ff_forward(sd, z)).max() < 2e-6
    # an injected dropout mask is the tile kernel's business
    x = rng.normal(size=(16384, I)).astype(np.float32)
    mask = ((rng.random((16384, H)) >= 0.2) / 0.8).astype(np.float32)
    ym = m(torch.from_numpy(x).cuda(), masks=torch.from_numpy(mask).cuda()).cpu().numpy()
    assert np.abs(ym - orc.ff_forward(sd, x, mask=mask)).max() < 2e-6


@pytest.mark.gpu
def test_imupose_on_the_cluster_kernel():
    """ImuPoseLSTM's 2 x 256 LSTM behind its input layer on ape_lstm_cluster<256, 2, 256, nmt> (AUTO) against the batch-tile
    kernel (same arithmetic up to the f32 summation order) and the oracle: one and two row tiles, a ragged last cluster,
    two launches (1024 rows > 16 clusters x 32), last-step and all-steps output"""
    from wear_mocap_ape_amd.estimate import nn_models
    sd = orc.make_imupose_state_dict(22, 14, 11)
    m = nn_models.ImuPoseLSTM(22, 256, 2, 14, device=0)
    m.load_state_dict(sd)
    rng = np.random.default_rng(5)
    # (round 5: above 512 windows -- where the old kernel needs a second launch -- the LSTM runs one layer per launch on lstm_upper32.hip's
    #  persistent clusters: fewer tiles than clusters, 32 tiles on 32 clusters, a ragged 33rd tile, two tiles per cluster + a ragged one,
    #  and two chunks of the 4096-window workspace)
    for B, T in ((5, 6), (16, 6), (40, 9), (300, 6), (512, 5), (513, 5), (600, 1), (700, 7), (1024, 2), (1024, 12), (1056, 5), (2090, 4), (4200, 3)):
        x = rng.normal(size=(B, T, 22)).astype(np.float32)
        xt = torch.from_numpy(x).cuda()
        want = "ape_lstm_upper32<32, true>" if B > 512 else "ape_lstm_cluster<256, 2, 256"
        assert want in m.set_kernel("auto").kernel_name(B, T)
        y_cl = m.set_kernel("auto")(xt, last_step_only=True).cpu().numpy()[:, 0]
        assert m.last_kernel() == ("ape_lstm_upper32" if B > 512 else "ape_lstm_cluster"), m.last_kernel()
        y_t16 = m.set_kernel("tile16")(xt, last_step_only=True).cpu().numpy()[:, 0]
        assert np.abs(y_cl - y_t16).max() < 2e-6, (B, T, float(np.abs(y_cl - y_t16).max()))
        sub = rng.choice(B, size=min(B, 24), replace=False)
        assert np.abs(y_cl[sub] - orc.imupose_forward(sd, x[sub])[:, -1]).max() < TOL_Y_SHORT
        if B <= 40:
            y_all = m.set_kernel("auto")(xt).cpu().numpy()
            assert np.abs(y_all - orc.imupose_forward(sd, x)).max() < TOL_Y_SHORT
            assert np.abs(y_all[:, -1] - y_cl).max() < 1e-6          # the all-steps head is a second launch
    m.set_kernel("auto")
    with pytest.raises(UserWarning):
        m.set_precision("f16")
    m.check()


# ---------------- aborted launches: staged with the test-hooks library, in a child process -----------------------------------
def test_abort_paths_on_the_test_hooks_library():
    """The product library has no entry point that can corrupt a handle: the tests that stage the state an aborted launch leaves
    behind (sticky status word, consumed tickets, the latency kernel's launch-number wrap) run tests/hooks/poke_cases.py in a
    CHILD process whose APE_HIP_LIB is lib/diag/libape_hip_testhooks.so = the product objects + ape_debug.hip (csrc/Makefile,
    target `hooks`).  One child at a time; the parent keeps no launch in flight meanwhile."""
    import os
    import subprocess
    import sys
    from tests.conftest import REPO
    lib = REPO / "arm-pose-estimation_amd" / "lib" / "diag" / "libape_hip_testhooks.so"
    assert lib.exists(), "make -C arm-pose-estimation_amd/csrc hooks"
    torch.cuda.synchronize()
    env = dict(os.environ, APE_HIP_LIB=str(lib))
    r = subprocess.run([sys.executable, "-m", "pytest", str(REPO / "tests" / "hooks" / "poke_cases.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=env, cwd=str(REPO), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_two_estimator_threads_on_the_latency_kernel(norm_stats):
    """The reference runs one Estimator per consumer thread (estimator.py:139-143).  Two threads, each with its own model handle on
    its own HIP stream, both on the cooperative latency kernel (B = 1), 2000 frames each: every frame must be the single-threaded
    result -- bit for bit while no launch gave up, within 1e-6 where ape_model_recover re-issued one on the batch-tile kernel
    (counted in ape_model_stats; no frame may be lost)."""
    import threading
    name, n_frames = "pocket", 2000
    models = [make_model(name, 7 + k, norm_stats[name])[0] for k in range(2)]
    cfg = orc.MODEL_CONFIGS[name]
    xs = [_synthetic_windows(norm_stats[name], 1, 6 + n_frames, cfg["I"], 50 + k)[0] for k in range(2)]      # a 50 Hz sequence per thread
    want = []
    for k in range(2):        # single-threaded pass: frame f = the window of rows f .. f+5
        w = np.stack([models[k](torch.from_numpy(np.ascontiguousarray(xs[k][f:f + 6][None])), last_step_only=True,
                                normalize_input=True).numpy()[0, 0] for f in range(n_frames)])
        want.append(w)
    got = [np.empty_like(want[0]), np.empty_like(want[1])]
    errs = []

    def worker(k):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for f in range(n_frames):
                    got[k][f] = models[k](torch.from_numpy(np.ascontiguousarray(xs[k][f:f + 6][None])), last_step_only=True,
                                          normalize_input=True).numpy()[0, 0]
        except Exception as exc:          # noqa: BLE001 -- reported below
            errs.append((k, repr(exc)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(2):
        st = models[k].stats()
        assert st["lost_calls"] == 0, st
        d = np.abs(got[k] - want[k]).max(axis=1)
        if st["reissued_calls"] == 0:
            assert np.array_equal(got[k], want[k]), (k, float(d.max()))
        else:
            assert float(d.max()) < 1e-6 and int((d > 0).sum()) <= st["reissued_calls"], (k, st, float(d.max()))


# ---------------- Monte-Carlo bank on the weight-stationary route (lstm_upper32.hip), shapes the other tests do not reach -----------
@pytest.mark.parametrize("name,S,n_mc,T", [("pocket", 83, 25, 1), ("watch", 83, 25, 2), ("pocket", 350, 7, 3), ("pocket", 41, 60, 6),
                                            ("watch", 1100, 2, 8),
                                            # round 5: the route from 513 sample rows on -- fewer tiles than clusters (every cluster on the
                                            # SOLO form), one-step windows there, a mix of one- and two-tile clusters
                                            ("pocket", 21, 25, 6), ("pocket", 27, 19, 1), ("watch", 60, 25, 8), ("pocket", 9, 60, 6)])
def test_mc_bank_cluster_route_against_batch_tile_route(golden, norm_stats, name, S, n_mc, T):
    """layer 0 once per stream + the layer above over the sample rows, both on the persistent cluster kernels (AUTO), against the
    same bank on the batch-tile kernels ('auto_gen1': same Philox counters, so the same dropout masks): window lengths 1 .. 8, sample
    rows from 513 on (the route's threshold), ragged last tiles, streams that straddle tiles, n_mc = 2 .. 60, several frames so
    that the window rings wrap.  Every stacked row's hand / elbow position to 5e-6 (float32 summation order), messages to 5e-5."""
    from wear_mocap_ape_amd.streams import StreamBank
    assert S * n_mc > 512
    stats = norm_stats[name]
    cfg = orc.MODEL_CONFIGS[name]
    feats = _synthetic_windows(stats, S, T + 3, cfg["I"], 77)
    outs = {}
    for kern in ("auto", "auto_gen1"):
        m, sd, _ = make_model(name, 21, stats)
        m.set_body(orc.DEFAULT_BODY)
        m.set_kernel(kern)
        bank = StreamBank(m, S, T, smooth=2, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=4242)
        res = []
        for f in range(T + 3):
            bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda())
            msg, tail = bank.step(with_tail=True)
            res.append((msg.cpu().numpy().copy(), tail.cpu().numpy().copy()))
        m.check()
        outs[kern] = res
        del bank
    # the opt-in plain in-XCD exchange form of the first-generation cluster kernel (APE_FLAG_IN_XCD_PLAIN; the bank kernels themselves
    # hand over write-through whatever the flag): the same bits as the default write-through form
    m, sd, _ = make_model(name, 21, stats)
    m.set_body(orc.DEFAULT_BODY)
    bank = StreamBank(m, S, T, smooth=2, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=4242)
    from wear_mocap_ape_amd import _hip
    bank._flags |= _hip.FLAG_IN_XCD_PLAIN
    for f in range(T + 3):
        bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda())
        msg, tail = bank.step(with_tail=True)
        assert np.array_equal(msg.cpu().numpy(), outs["auto"][f][0]) and np.array_equal(tail.cpu().numpy(), outs["auto"][f][1])
    m.check()
    del bank
    worst_tail = max(float(np.abs(a[1] - b[1]).max()) for a, b in zip(outs["auto"], outs["auto_gen1"]))
    worst_msg = max(float(np.abs(a[0] - b[0]).max()) for a, b in zip(outs["auto"], outs["auto_gen1"]))
    assert worst_tail < 5e-6 and worst_msg < 5e-5, (worst_tail, worst_msg)
    assert worst_tail > 0.0          # (two different kernels: not the same launch twice)
