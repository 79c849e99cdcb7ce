"""Every in-launch hand-over of the library under UNEVEN load, against the oracle (MI355X guide, visibility section:
"test every hand-off under uneven load, consumer L1-warm, checking every word: idle chips, uniform load and L1-cold consumers hide
these failures").  While a cooperative kernel runs on the test's stream, a second stream keeps a queue of large device-to-device copies
going: the copies take whatever CUs are free (the kernel's workgroups start staggered behind them), stream through every XCD's L2 and
the memory side for as long as the kernel runs, and on shapes that fill only part of the chip they run BESIDE it throughout.  The
consumers are L1-warm by construction (every kernel re-reads its exchange slots every second step).  Every output row is compared with
the oracle; an aborted launch (bounded spin) would be re-issued on the batch-tile kernel by the mirror, so the cases also assert that
nothing aborted and name the kernel that ran.

Runs in a child process on lib/diag/libape_hip_testhooks.so (= the product objects + the injected-mask hook the two bank routes need):
tests/test_hip_round5.py::test_hand_overs_under_uneven_load."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ape_oracle as orc
from tests.test_hip_parity import make_model, _synthetic_windows

pytestmark = pytest.mark.gpu


class Streamer:
    """a queue of 256 MiB device copies on a second stream: ~80 us each at the chip's copy rate, `n` of them per burst"""

    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.a = torch.full((64 << 20,), 1.0, dtype=torch.float32, device="cuda")
        self.b = torch.empty_like(self.a)
        self.bursts = 0

    def burst(self, n=24):
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                self.b.copy_(self.a, non_blocking=True)
        self.bursts += 1

    def drain(self):
        self.stream.synchronize()


@pytest.fixture(scope="module")
def streamer():
    s = Streamer()
    yield s
    s.drain()
    assert s.bursts > 0 and float(s.b[12345]) == 1.0


def _no_abort(m):
    m.check()
    assert m.stats()["aborted_checks"] == 0, m.stats()


@pytest.mark.parametrize("name,B,T,prec,kernel,expect", [
    ("pocket", 640, 16, "f32", "cluster", "ape_lstm_cluster32"),         # 24 clusters of 8: three quarters of the chip, copies beside it
    ("pocket", 1024, 64, "f32", "cluster", "ape_lstm_cluster32"),        # the benchmark shape (whole chip: staggered start, loaded memory side)
    ("pocket", 1024, 6, "f32", "cluster", "ape_lstm_cluster32"),         # the short-window instantiation (end forms)
    ("watch", 600, 8, "f32", "cluster", "ape_lstm_cluster32"),
    ("pocket", 512, 8, "f32", "cluster_gen1", "ape_lstm_cluster"),       # first generation, clusters within block-index classes
    ("pocket", 300, 6, "f32", "cluster_gen1", "ape_lstm_cluster"),
    ("uarm", 1024, 6, "f32", "cluster_gen1", "ape_lstm_cluster"),
    ("uarm", 700, 50, "f32", "cluster", "ape_lstm_cluster16"),
    ("uarm", 1024, 64, "f32", "cluster", "ape_lstm_cluster16"),
    ("uarm", 1024, 6, "f32", "auto", "ape_lstm_level16"),                # round 6: tagged granules polled by every wave, two agents per workgroup
    ("uarm", 700, 13, "f32", "auto", "ape_lstm_level16"),                # ... with row tiles and whole clusters past the end of the batch
    ("uarm", 530, 2, "f32", "auto", "ape_lstm_level16"),                 # ... and a window shorter than the model is deep
    ("uarm", 300, 6, "f32", "auto", "ape_lstm_level16"),                 # ... one row tile per cluster (the idle agent of every workgroup), copies beside it
    ("watch", 700, 8, "f16", "cluster", "ape_lstm_cluster_f16v2"),
    ("watch", 1024, 64, "f16", "cluster", "ape_lstm_cluster_f16v2"),
    ("pocket", 200, 6, "f16_gen1", "cluster", "ape_lstm_cluster_f16"),
    ("pocket", 1, 6, "f32", "auto", "ape_lstm_cluster_small"),            # the latency kernel (tagged granules)
    ("pocket", 4, 6, "f32", "auto", "ape_lstm_cluster_small"),
])
def test_lstm_kernels_under_uneven_load(norm_stats, streamer, name, B, T, prec, kernel, expect):
    st = norm_stats[name]
    m, sd, cfg = make_model(name, 13, st)
    x = _synthetic_windows(st, B, T, cfg["I"], B + T)
    xd = torch.from_numpy(x).cuda()
    xn = ((x.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
    if prec != "f32": m.set_precision(prec)
    m.set_kernel(kernel)
    ref = orc.lstm_forward(sd, xn, storage="f16")[:, -1] if prec != "f32" else orc.lstm_forward(sd, xn)[:, -1]
    tol = 3e-4 if prec != "f32" else 1e-6
    worst = 0.0
    for rep in range(4):
        streamer.burst()
        y = m(xd, last_step_only=True, normalize_input=True)           # launched while the copies run
        assert expect in m.last_kernel(), m.last_kernel()
        worst = max(worst, float(np.abs(y.cpu().numpy()[:, 0] - ref).max()))
    streamer.drain()
    _no_abort(m)
    assert worst < tol, (name, B, T, prec, worst)


def test_mlp_pipeline_under_uneven_load(streamer):
    from wear_mocap_ape_amd.estimate import nn_models
    I, H, O = 22, 256, 14
    sd = orc.make_ff_state_dict(I, H, 2, O, 9)
    m = nn_models.DropoutFF(output_size=O, hidden_layer_size=H, hidden_layer_count=2, input_size=I, dropout=0.2, device=0)
    m.load_state_dict(sd)
    x = np.random.default_rng(5).normal(size=(40000, I)).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    ref = orc.ff_forward(sd, x)
    for rep in range(4):
        streamer.burst()
        y = m(xt).cpu().numpy()
        assert np.abs(y - ref).max() < 2e-6, float(np.abs(y - ref).max())
    streamer.drain()
    m.check()


@pytest.mark.parametrize("name,S,n_mc,route", [("pocket", 170, 25, "ape_lstm_upper32"), ("watch", 100, 25, "ape_lstm_upper32"),
                                               ("pocket", 30, 25, "ape_lstm_upper32"), ("watch", 50, 25, "ape_lstm_upper32"),      # (one-tile clusters: SOLO form)
                                               ("uarm", 100, 50, "ape_lstm_upper128"), ("uarm", 160, 25, "ape_lstm_upper128"), ("uarm", 30, 50, "ape_lstm_upper128"),
                                               ("pocket", 1, 25, "ape_lstm_mc_small")])
def test_bank_routes_under_uneven_load(norm_stats, streamer, name, S, n_mc, route):
    """the Monte-Carlo banks' weight-stationary routes (and the one-stream latency kernel) with injected masks: every sample row of every
    frame against the oracle's masked cell loop while the copies run"""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    lib = _hip.lib()
    assert hasattr(lib, "ape_debug_set_bank_masks"), "this file runs on the test-hooks library (APE_HIP_LIB)"
    lib.ape_debug_set_bank_masks.restype, lib.ape_debug_set_bank_masks.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    lib.ape_debug_bank_targets.restype, lib.ape_debug_bank_targets.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    st = norm_stats[name]
    m, sd, cfg = make_model(name, 5, st)
    m.set_body(orc.DEFAULT_BODY)
    T, I, O, H, L = cfg["T"], cfg["I"], cfg["O"], cfg["H"], cfg["L"]
    rows = S * n_mc
    rng = np.random.default_rng(S * 1000 + n_mc)
    feats = _synthetic_windows(st, S, T + 1, I, 8)
    bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=3)
    hist = [[] for _ in range(S)]
    for f in range(T + 1):
        masks = [(rng.random((rows, T, H)) >= 0.2).astype(np.float32) / np.float32(0.8) for _ in range(L - 1)]
        md = torch.from_numpy(np.stack(masks)).cuda()
        assert lib.ape_debug_set_bank_masks(bank._handle, C.c_void_p(md.data_ptr())) == 0
        bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda())
        torch.cuda.synchronize()
        streamer.burst()
        bank.step()
        assert m.last_kernel() == route, m.last_kernel()
        y = np.empty((rows, O), dtype=np.float32)
        assert lib.ape_debug_bank_targets(bank._handle, y.ctypes.data_as(C.c_void_p)) == 0
        wins = []
        for s in range(S):
            hist[s].append(feats[s, f])
            while len(hist[s]) < T:
                hist[s].append(feats[s, f])
            del hist[s][:len(hist[s]) - T]
            xn = ((np.stack(hist[s]).astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
            wins.append(np.repeat(xn[None], n_mc, axis=0))
        ref = orc.lstm_forward(sd, np.concatenate(wins), masks=masks)[:, -1, :]
        assert np.abs(y - ref).max() < 1e-6, (f, float(np.abs(y - ref).max()))
    streamer.drain()
    assert lib.ape_debug_set_bank_masks(bank._handle, None) == 0
    _no_abort(m)


def test_fresh_banks_layer0_sequence_under_uneven_load(streamer):
    """the layer-0 form of lstm_upper32.hip publishes every step's slices into the sequence launch B reads: fresh banks (first launches on
    untouched buffers), push and step back to back with no host synchronisation in between, copies beside them -- and EVERY word of the
    sequence against a float64 layer 0 in numpy.  (Round 5: once in ~2000 such frames sixteen lanes of one wave published the integer
    slice epoch in place of a hidden value -- an asm store's data registers re-used inside its two wait states, async_look.h; the static
    scan of tests/test_asm_hazards.py is the guard, this is the observation it was found by.)"""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.estimate import nn_models
    from wear_mocap_ape_amd.streams import StreamBank
    lib = _hip.lib()
    assert hasattr(lib, "ape_debug_bank_buffer"), "this file runs on the test-hooks library (APE_HIP_LIB)"
    lib.ape_debug_bank_buffer.restype = C.c_int
    lib.ape_debug_bank_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    cfg = orc.MODEL_CONFIGS["watch"]
    T, I, O, H, S, n_mc = cfg["T"], cfg["I"], cfg["O"], cfg["H"], 130, 20      # (above 96 streams: launch A on the SEQ form of lstm_upper32.hip)
    tiles = (S + 31) // 32
    rng = np.random.default_rng(41)
    sg = lambda v: 1.0 / (1.0 + np.exp(-v))
    for bank_no in range(100):
        sd = orc.make_state_dict(I, H, cfg["L"], O, int(rng.integers(100)))
        m = nn_models.DropoutLSTM(I, H, cfg["L"], O, dropout=0.2, device=0); m.load_state_dict(sd); m.set_body(orc.DEFAULT_BODY)
        bank = StreamBank(m, S, T, smooth=2, normalize=False, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=bank_no)
        w_ih, w_hh = sd["lstm.weight_ih_l0"].astype(np.float64), sd["lstm.weight_hh_l0"].astype(np.float64)
        b = sd["lstm.bias_ih_l0"].astype(np.float64) + sd["lstm.bias_hh_l0"].astype(np.float64)
        hist = None
        for f in range(2):
            xx = rng.normal(size=(S, I)).astype(np.float32)
            streamer.burst(8)
            bank.push_features(torch.from_numpy(xx).cuda())
            bank.step()
            assert m.last_kernel() == "ape_lstm_upper32", m.last_kernel()
            hist = np.repeat(xx[:, None], T, axis=1) if hist is None else np.concatenate([hist[:, 1:], xx[:, None]], axis=1)
            got = np.empty(tiles * T * 8192, dtype=np.float32); n = C.c_size_t(0)
            assert lib.ape_debug_bank_buffer(bank._handle, 2, got.ctypes.data_as(C.c_void_p), got.nbytes, C.byref(n)) == 0 and n.value == got.nbytes
            h = np.zeros((S, H)); c = np.zeros_like(h); want = np.zeros((tiles * 32, T, H))
            for t in range(T):
                pre = hist[:, t].astype(np.float64) @ w_ih.T + h @ w_hh.T + b
                c = sg(pre[:, H:2 * H]) * c + sg(pre[:, :H]) * np.tanh(pre[:, 2 * H:3 * H])
                h = sg(pre[:, 3 * H:]) * np.tanh(c)
                want[:S, t] = h
            want = want.reshape(tiles, 32, T, 32, 8).transpose(0, 2, 3, 1, 4)           # [tile][step][k-block][row][8 units]
            d = np.abs(got.reshape(want.shape) - want)
            d[tiles - 1, :, :, S - 32 * (tiles - 1):] = 0.0                                # rows past the bank: computed, never read
            assert d.max() < 2e-6, (bank_no, f, float(d.max()), np.argwhere(d > 2e-6)[:8].tolist())
        _no_abort(m)
        del bank, m
    streamer.drain()


@pytest.mark.parametrize("B,T", [(1024, 9), (1500, 5), (2090, 4)])
def test_imupose_layer_split_under_uneven_load(streamer, B, T):
    """ImuPoseLSTM from 1024 windows on: one layer per launch on lstm_upper32.hip's clusters -- one tile per cluster (the solo form: the
    own gather under the input span), a mix of one- and two-tile clusters, two tiles everywhere + a ragged one; every window against the
    oracle while the copies run, and once more on the blocking form of the one-tile clusters (ALT_FORM), bit for bit the same arithmetic"""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.estimate import nn_models
    sd = orc.make_imupose_state_dict(22, 14, 21)
    m = nn_models.ImuPoseLSTM(22, 256, 2, 14, device=0)
    m.load_state_dict(sd)
    x = np.random.default_rng(B + T).normal(size=(B, T, 22)).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    ref = orc.imupose_forward(sd, x)[:, -1]
    lib = _hip.lib()
    ys = {}
    for flags in (0, _hip.FLAG_ALT_FORM):
        y = torch.empty((B, 14), dtype=torch.float32, device="cuda")
        worst = 0.0
        for rep in range(4):
            streamer.burst()
            _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(xt.data_ptr()), B, T, flags, None, 0.0, 0, C.c_void_p(y.data_ptr()), None), "fwd")
            assert m.last_kernel() == "ape_lstm_upper32", m.last_kernel()
            worst = max(worst, float(np.abs(y.cpu().numpy() - ref).max()))
        assert worst < 2e-6, (B, T, flags, worst)
        ys[flags] = y.cpu().numpy()
    streamer.drain()
    _no_abort(m)
    assert np.array_equal(ys[0], ys[_hip.FLAG_ALT_FORM])
