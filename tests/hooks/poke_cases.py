"""Abort paths of the weight-stationary kernels, staged with `ape_debug_poke` -- an entry point of lib/diag/libape_hip_testhooks.so
only (the product objects + csrc/ape_debug.hip).  Not collected with the suite: tests/test_hip_round2.py runs this file in a child
process whose APE_HIP_LIB names that library (`test_abort_paths_on_the_test_hooks_library`)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ape_oracle as orc
from tests.test_hip_parity import make_model, _synthetic_windows, TOL_Y_SHORT

pytestmark = pytest.mark.gpu


def _poke(lib):
    assert hasattr(lib, "ape_debug_poke"), "this file runs on the test-hooks library (APE_HIP_LIB)"
    lib.ape_debug_poke.restype, lib.ape_debug_poke.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_uint]
    return lib.ape_debug_poke


@pytest.mark.parametrize("name,B", [("pocket", 1), ("pocket", 64), ("pocket", 1024), ("uarm", 1024), ("uarm", 1000)])
def test_aborted_cluster_launch_is_reissued_or_reported(norm_stats, name, B):
    """state an aborted launch leaves behind (sticky status word set, tickets consumed): the next call on that handle must never
    return garbage silently.  Host outputs: the Python mirror recovers (ape_model_recover re-issues the call on the batch-tile
    kernel: the frame is NOT lost, the result is within 1e-6 of the cooperative kernel's, ape_model_stats counts it).  Device
    outputs: the strict check raises and resets; `recover()` instead re-issues.  Afterwards the handle works again, bit-equal."""
    from wear_mocap_ape_amd import _hip
    m, sd, cfg = make_model(name, 0, norm_stats[name])
    poke = _poke(_hip.lib())
    T = cfg["T"] if B < 1024 else 64
    if B == 1024: assert m.kernel_name(B, T) == {"pocket": "ape_lstm_cluster32<256, 2, 32, false>", "uarm": "ape_lstm_cluster16<128, 3, 64, 2>"}[name]
    if B == 1000: assert m.kernel_name(B, T) == "ape_lstm_level16<128, 3, 64>"      # (round 6: tagged granules, no flags -- the same status / ticket protocol)
    x = torch.from_numpy(_synthetic_windows(norm_stats[name], B, T, cfg["I"], 3))
    good = m(x, last_step_only=True, normalize_input=True).numpy().copy()
    m.set_kernel("tile16")
    tile = m(x, last_step_only=True, normalize_input=True).numpy().copy()
    m.set_kernel("auto")
    assert np.abs(tile - good).max() < 1e-6
    reissued = aborted = lost = 0
    # status word set / tickets beyond any grid (the one-cluster latency kernel at B = 1 takes no tickets; the others draw theirs per
    # block-index class: word 6 = the class-0 ticket)
    for which, value in (((0, 1),) if B == 1 else ((0, 1), (6, 100000))):
        assert poke(m.handle, which, value) == 0
        out = m(x, last_step_only=True, normalize_input=True).numpy()      # host output: recovered before it is handed out
        reissued += 1; aborted += 1
        assert np.array_equal(out, tile), "the re-issue runs on the batch-tile kernel"
        assert m.stats() == {"aborted_checks": aborted, "reissued_calls": reissued, "lost_calls": lost}
        assert np.array_equal(m(x, last_step_only=True, normalize_input=True).numpy(), good)
        # device output + strict check: loud, resets, the calls are counted as lost
        assert poke(m.handle, which, value) == 0
        m(x.cuda(), last_step_only=True, normalize_input=True)
        with pytest.raises(UserWarning, match="aborted"):
            m.check()
        m.check()
        aborted += 1; lost += 1
        assert m.stats() == {"aborted_checks": aborted, "reissued_calls": reissued, "lost_calls": lost}
        # device output + recover: re-issued in place
        assert poke(m.handle, which, value) == 0
        xd = x.cuda()
        yd = m(xd, last_step_only=True, normalize_input=True)
        m.recover()
        reissued += 1; aborted += 1
        assert np.array_equal(yd.cpu().numpy(), tile)
        assert m.stats() == {"aborted_checks": aborted, "reissued_calls": reissued, "lost_calls": lost}
        assert np.array_equal(m(x, last_step_only=True, normalize_input=True).numpy(), good)


def test_aborted_imupose_layer_split_is_reissued():
    """ImuPoseLSTM's layer-split route (two launches of lstm_upper32.hip per call, round 5) with the state an aborted launch leaves behind: the
    host-output call is re-issued on the batch-tile kernel, the strict check of a device-output call raises and resets, recover() re-issues
    in place; afterwards the handle gives the route's own bits again"""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.estimate import nn_models
    sd = orc.make_imupose_state_dict(22, 14, 4)
    m = nn_models.ImuPoseLSTM(22, 256, 2, 14, device=0)
    m.load_state_dict(sd)
    poke = _poke(_hip.lib())
    B, T = 1024, 5
    assert "ape_lstm_upper32<32, true>" in m.kernel_name(B, T)
    x = torch.from_numpy(np.random.default_rng(2).normal(size=(B, T, 22)).astype(np.float32))
    good = m(x, last_step_only=True).numpy().copy()
    assert m.last_kernel() == "ape_lstm_upper32"
    tile = m.set_kernel("tile16")(x, last_step_only=True).numpy().copy()
    m.set_kernel("auto")
    assert np.abs(tile - good).max() < 2e-6
    assert np.abs(good[:, 0] - orc.imupose_forward(sd, x.numpy())[:, -1]).max() < 2e-6
    reissued = aborted = lost = 0
    for which, value in ((0, 1), (6, 100000)):
        assert poke(m.handle, which, value) == 0
        out = m(x, last_step_only=True).numpy()
        reissued += 1; aborted += 1
        assert np.array_equal(out, tile), "the re-issue runs on the batch-tile kernel"
        assert m.stats() == {"aborted_checks": aborted, "reissued_calls": reissued, "lost_calls": lost}
        assert np.array_equal(m(x, last_step_only=True).numpy(), good)
        assert poke(m.handle, which, value) == 0
        m(x.cuda(), last_step_only=True)
        with pytest.raises(UserWarning, match="aborted"):
            m.check()
        m.check()
        aborted += 1; lost += 1
        assert poke(m.handle, which, value) == 0
        yd = m(x.cuda(), last_step_only=True)
        m.recover()
        reissued += 1; aborted += 1
        assert np.array_equal(yd.cpu().numpy(), tile)
        assert m.stats() == {"aborted_checks": aborted, "reissued_calls": reissued, "lost_calls": lost}
        assert np.array_equal(m(x, last_step_only=True).numpy(), good)


@pytest.mark.parametrize("name,S,n_mc,route", [("pocket", 330, 25, "ape_lstm_upper32"), ("uarm", 170, 50, "ape_lstm_upper128")])
def test_aborted_infer_and_bank_step_are_reissued(norm_stats, name, S, n_mc, route):
    """ape_infer (LSTM + post-filter) and a Monte-Carlo stream-bank step behind an aborted launch: recover re-issues both; a bank
    that has moved on since the aborted step cannot be re-issued and says so.  Both weight-stationary bank routes (2 x 256: lstm_upper32.hip,
    3 x 128: lstm_upper128.hip)."""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    m, sd, cfg = make_model(name, 2, norm_stats[name])
    m.set_body(orc.DEFAULT_BODY)
    lib = _hip.lib()
    poke = _poke(lib)
    B, T = 700, cfg["T"]
    x = torch.from_numpy(_synthetic_windows(norm_stats[name], B, T, cfg["I"], 5)).cuda()
    est = torch.empty((B, 21 if name == "pocket" else 14), dtype=torch.float64, device="cuda")

    def infer():
        _hip.check(lib.ape_infer(m.handle, C.c_void_p(x.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT, None, C.c_void_p(est.data_ptr()),
                                 _hip.F64, None), "ape_infer")
    infer(); m.check()
    good = est.cpu().numpy().copy()
    assert poke(m.handle, 0, 1) == 0
    est.zero_()
    infer()
    m.recover()
    assert np.abs(est.cpu().numpy() - good).max() < 2e-6
    assert m.stats()["reissued_calls"] == 1
    # a Monte-Carlo bank on the weight-stationary upper-layer kernel: the same step again on the batch-tile route (same Philox masks)
    feats = _synthetic_windows(norm_stats[name], S, 4, cfg["I"], 6)
    def run(abort_at):
        bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=99)
        outs = []
        for f in range(4):
            bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda())
            if f == abort_at:
                assert poke(m.handle, 0, 1) == 0
            msg, tail = bank.step(with_tail=True)
            assert m.last_kernel() == route
            bank.recover()
            outs.append((msg.cpu().numpy().copy(), tail.cpu().numpy().copy()))
        return outs
    ref, got = run(-1), run(2)
    # (frame 2 of `got` ran on the batch-tile route: same Philox masks, float32 summation order of another kernel; the budget is
    #  SURVEY 8d's for quaternions / origins behind a float32 regressor, 5e-5 -- the other frames are bit-equal)
    for f, ((a, b), (c, d)) in enumerate(zip(ref, got)):
        if f == 2:
            assert np.abs(a - c).max() < 5e-5 and np.abs(b - d).max() < 5e-5
            assert np.abs(a - c).max() > 0.0
        else:
            assert np.array_equal(a, c) and np.array_equal(b, d)
    assert m.stats()["reissued_calls"] == 2 and m.stats()["lost_calls"] == 0
    # moved on: a row pushed behind the aborted step
    bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=99)
    bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, 0])).cuda())
    assert poke(m.handle, 0, 1) == 0
    bank.step()
    bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, 1])).cuda())
    with pytest.raises(UserWarning, match="could not be re-issued"):
        bank.recover()
    assert m.stats()["lost_calls"] == 1
    m.check()


@pytest.mark.gpu
def test_mlp_pipeline_abort_and_graph_replay():
    """the pipeline kernel is loud and recoverable like the cluster kernels (a sticky status word / class tickets an aborted launch
    left behind: host results raise, `check()` raises once, then the handle works again bit for bit), and a captured launch replays
    on new data (the kernel re-zeroes its own hand-over words)"""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.estimate import nn_models
    I, H, O, N = 22, 256, 14, 20000
    m = nn_models.DropoutFF(output_size=O, hidden_layer_size=H, hidden_layer_count=2, input_size=I, dropout=0.2, device=0)
    m.load_state_dict(orc.make_ff_state_dict(I, H, 2, O, 4))
    lib = _hip.lib()
    poke = _poke(lib)
    assert m.kernel_name(N, 1) == "ape_mlp_pipe"
    rng = np.random.default_rng(2)
    x = rng.normal(size=(N, I)).astype(np.float32)
    good = m(x).numpy().copy()
    for which, value in ((4, 1), (5, 100000)):
        assert poke(m.handle, which, value) == 0
        out = m(x).numpy()                                  # host output: re-issued on the tile kernel before it is handed out
        assert np.abs(out - good).max() < 1e-6
        assert np.array_equal(m(x).numpy(), good)
        assert poke(m.handle, which, value) == 0
        m(torch.from_numpy(x).cuda())                       # device output: the caller checks
        with pytest.raises(UserWarning, match="aborted"):
            m.check()
        m.check()
        assert np.array_equal(m(x).numpy(), good)
    # graph capture and replay
    xin = torch.from_numpy(x).cuda()[:, None, :].contiguous()
    x2 = torch.from_numpy(rng.normal(size=(N, 1, I)).astype(np.float32)).cuda()
    y_graph = torch.zeros((N, O), device="cuda")
    y_eager = torch.zeros((N, O), device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call = lambda src, out, stream: _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(src.data_ptr()), N, 1, 0, None, 0.0, 0,
                                                                        C.c_void_p(out.data_ptr()), stream), "fwd")
        call(xin, y_graph, C.c_void_p(side.cuda_stream))
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            call(xin, y_graph, C.c_void_p(side.cuda_stream))
    torch.cuda.current_stream().wait_stream(side)
    first = xin.clone()
    for data in (first, x2, first):
        xin.copy_(data)
        graph.replay()
        torch.cuda.synchronize()
        call(data, y_eager, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert torch.equal(y_graph, y_eager)
    m.check()
    assert np.array_equal(y_eager.cpu().numpy(), good)


# ---------------- ImuPoseLSTM on the weight-stationary cluster kernel (256-wide layer-0 input) ---------------------------


# ---------------- the latency kernel's launch number wraps after 2^20 launches ----------------------------------------------
@pytest.mark.gpu
def test_latency_kernel_launch_number_wrap():
    """the granule tags carry a 20-bit launch number kept on the device; at the wrap the last member out zeroes the granules, so a
    tag of 2^20 launches ago can never be taken for a fresh one: launches across the wrap give the same bits as before it"""
    from wear_mocap_ape_amd import _hip
    model, sd, cfg = make_model("pocket", 14)
    lib = _hip.lib()
    poke = _poke(lib)
    rng = np.random.default_rng(8)
    xs = [torch.from_numpy(rng.normal(size=(B, 6, cfg["I"])).astype(np.float32)).cuda() for B in (1, 3, 1, 2, 4, 1)]
    before = [model(x, last_step_only=True).cpu().numpy() for x in xs]
    for b, x in zip(before, xs):
        assert np.abs(b[:, 0] - orc.lstm_forward(sd, x.cpu().numpy())[:, -1]).max() < TOL_Y_SHORT
    assert poke(model.handle, 3, 0xFFFFD) == 0
    for rep in range(2):                            # launches 0xFFFFD, E, F (wrap: granules zeroed), 0, 1, 2, ...
        for b, x in zip(before, xs):
            assert np.array_equal(model(x, last_step_only=True).cpu().numpy(), b)
    model.check()


def test_level_kernel_launch_number_wrap(norm_stats):
    """lstm_level16.hip's granule tags = (20-bit launch number << 12) | level, the number kept on the device and bumped by the last workgroup
    out; at the wrap that workgroup zeroes the granule buffer (tag 0 is never awaited), so a tag of 2^20 launches ago cannot pass for a
    fresh one: launches across the wrap -- both forms, one and two row tiles per cluster -- give the same bits as before it"""
    from wear_mocap_ape_amd import _hip
    st = norm_stats["uarm"]
    model, sd, cfg = make_model("uarm", 14, st)
    poke = _poke(_hip.lib())
    xs = [torch.from_numpy(_synthetic_windows(st, B, T, cfg["I"], B)).cuda() for B, T in ((1024, 6), (40, 6), (700, 3), (512, 9))]
    before = []
    for x in xs:
        before.append(model(x, last_step_only=True, normalize_input=True).cpu().numpy())
        assert model.last_kernel() == "ape_lstm_level16"
        xn = ((x.cpu().numpy().astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
        assert np.abs(before[-1][:, 0] - orc.lstm_forward(sd, xn)[:, -1]).max() < 1e-6
    assert poke(model.handle, 8, 0xFFFFD) == 0
    for rep in range(2):                            # launches 0xFFFFD, E, F (wrap: granules zeroed), 0, 1, 2, ...
        for b, x in zip(before, xs):
            assert np.array_equal(model(x, last_step_only=True, normalize_input=True).cpu().numpy(), b)
    model.check()


@pytest.mark.parametrize("name,S,n_mc", [("pocket", 330, 25), ("uarm", 170, 50)])
def test_mc_bank_in_several_chunks(norm_stats, name, S, n_mc):
    """the weight-stationary route handles the sample rows in chunks (one launch each, <= 2 GiB of pre-laid input): a bank cut
    into chunks of 2048 rows by the test hook must give the bits of the same bank in one chunk (Philox counters and tile
    contents are functions of the GLOBAL row index)"""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    # (8250 / 8500 sample rows: 4 chunks of 2048 + a ragged one)
    stats = norm_stats[name]
    cfg = orc.MODEL_CONFIGS[name]
    lib = _hip.lib()
    lib.ape_debug_set_chunk_rows.restype, lib.ape_debug_set_chunk_rows.argtypes = C.c_int, [C.c_void_p, C.c_int]
    feats = _synthetic_windows(stats, S, 8, cfg["I"], 91)
    outs = []
    for chunk in (None, 2048):
        m, sd, _ = make_model(name, 3, stats)
        m.set_body(orc.DEFAULT_BODY)
        bank = StreamBank(m, S, cfg["T"], smooth=1, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=7)
        if chunk:
            assert lib.ape_debug_set_chunk_rows(bank._handle, chunk) == 0
        res = []
        for f in range(8):
            bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda())
            msg, tail = bank.step(with_tail=True)
            res.append((msg.cpu().numpy().copy(), tail.cpu().numpy().copy()))
        m.check()
        outs.append(res)
        del bank
    for (a, b), (c, d) in zip(*outs):
        assert np.array_equal(a, c) and np.array_equal(b, d)


def test_recover_reissues_a_chain_and_reports_what_it_cannot(norm_stats):
    """the journal of a handle: (1) a chain forward -> fk -> msg_reduce behind an aborted launch is re-issued in order and ends
    with the good results; (2) more than 64 pending calls cannot be replayed: loud, counted as lost, handle usable; (3) a bank
    destroyed with a pending step takes the step out of the journal (lost, not a dangling pointer)."""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    name = "pocket"
    m, sd, cfg = make_model(name, 9, norm_stats[name])
    m.set_body(orc.DEFAULT_BODY)
    lib = _hip.lib()
    poke = _poke(lib)
    B, T, O = 300, cfg["T"], cfg["O"]
    x = torch.from_numpy(_synthetic_windows(norm_stats[name], B, T, cfg["I"], 12)).cuda()
    y = torch.empty((B, O), dtype=torch.float32, device="cuda")
    est = torch.empty((B, 21), dtype=torch.float64, device="cuda")
    msg = torch.empty((25,), dtype=torch.float64, device="cuda")

    def chain():
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT, None, 0.0, 0,
                                        C.c_void_p(y.data_ptr()), None), "fwd")
        _hip.check(lib.ape_fk(m.handle, C.c_void_p(y.data_ptr()), _hip.F32, B, 1, C.c_void_p(est.data_ptr()), _hip.F64, None), "fk")
        _hip.check(lib.ape_msg_reduce(m.handle, C.c_void_p(est.data_ptr()), B, C.c_void_p(msg.data_ptr()), None), "msg")
    chain(); m.check()
    good = (y.cpu().numpy().copy(), est.cpu().numpy().copy(), msg.cpu().numpy().copy())
    assert poke(m.handle, 0, 1) == 0
    y.zero_(); est.zero_(); msg.zero_()
    chain()
    m.recover()
    assert m.stats() == {"aborted_checks": 1, "reissued_calls": 3, "lost_calls": 0}
    assert np.abs(y.cpu().numpy() - good[0]).max() < 1e-6
    assert np.abs(est.cpu().numpy() - good[1]).max() < 5e-5 and np.abs(msg.cpu().numpy() - good[2]).max() < 5e-5
    # (2) journal overflow
    assert poke(m.handle, 0, 1) == 0
    for _ in range(70):
        _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT, None, 0.0, 0,
                                        C.c_void_p(y.data_ptr()), None), "fwd")
    with pytest.raises(UserWarning, match="more than 64 calls"):
        m.recover()
    st = m.stats()
    assert st["aborted_checks"] == 2 and st["reissued_calls"] == 3 and st["lost_calls"] == 65, st
    chain(); m.recover()
    assert np.array_equal(y.cpu().numpy(), good[0])
    # (3) a bank that is gone
    S, n_mc = 100, 25
    bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc, dropout=0.2, seed=5)
    bank.push_features(torch.from_numpy(_synthetic_windows(norm_stats[name], S, 1, cfg["I"], 13)[:, 0].copy()).cuda())
    assert poke(m.handle, 0, 1) == 0
    bank.step()
    del bank
    with pytest.raises(UserWarning, match="could not be re-issued"):
        m.recover()
    m.check()
    chain(); m.recover()
    assert np.array_equal(y.cpu().numpy(), good[0])


def test_mc_latency_kernel_abort_recover_and_launch_number_wrap(norm_stats, golden, tmp_path, monkeypatch):
    """the Monte-Carlo latency kernel (lstm_mc_small.hip) behind the state an aborted launch leaves: a host-output call is re-issued
    on the batch-tile kernel under the same Philox counters (same samples, 1e-6); the estimators' device-resident frame
    (ape_streams_frame_host) recovers inside the call; the 20-bit launch number of its granule tags wraps without a stale tag
    being taken for a fresh one."""
    from array import array
    from wear_mocap_ape_amd import _hip, config
    from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    from tests.test_hip_parity import _deploy_dir
    name = "pocket"
    m, sd, cfg = make_model(name, 0, norm_stats[name])
    poke = _poke(_hip.lib())
    x = torch.from_numpy(_synthetic_windows(norm_stats[name], 1, cfg["T"], cfg["I"], 3))

    def sample(n=25):
        m.manual_seed(5)
        return m.monte_carlo_predictions(n, x, last_step_only=True).numpy().copy()
    good = sample()
    assert m.last_kernel() == "ape_lstm_mc_small"
    m.set_kernel("tile16")
    tile = sample()
    m.set_kernel("auto")
    assert np.abs(tile - good).max() < 1e-6 and np.abs(good - good[0]).max() > 1e-3
    assert poke(m.handle, 0, 1) == 0
    out = sample()                                   # host output: recovered before it is handed out
    assert np.array_equal(out, tile), "the re-issue runs on the batch-tile kernel with the same masks"
    assert m.stats() == {"aborted_checks": 1, "reissued_calls": 1, "lost_calls": 0}
    assert np.array_equal(sample(), good)
    # launch-number wrap
    big = sample(100)
    assert poke(m.handle, 7, 0xFFFFD) == 0
    for rep in range(6):
        assert np.array_equal(sample(), good) and np.array_equal(sample(100), big)
    m.check()
    # the consumer loop's frame: recovery is part of ape_streams_frame_host
    g = golden("stream_trace_pocket.npz")
    deploy, h = _deploy_dir(tmp_path, name, int(g["weights_seed"]), dropout=0.0)
    monkeypatch.setitem(config.PATHS, "deploy", deploy)
    est = WatchPhonePocketNN(model_hash=h, smooth=3, add_mc_samples=True, monte_carlo_samples=4)
    for f, row32 in enumerate(g["rows"][:10]):
        if f in (2, 7):
            assert poke(est._hip_model().handle, 0, 1) == 0
        msg = est.process_row(array("f", row32.tolist()))
        assert np.abs(np.asarray(msg) - g["msg_s3_mc4"][f]).max() < 5e-6, f
    st = est._hip_model().stats()
    assert st["aborted_checks"] == 2 and st["reissued_calls"] == 2 and st["lost_calls"] == 0, st
    # ... also with dropout on (the latency kernel aborts, the batch-tile route draws the same samples)
    deploy, h = _deploy_dir(tmp_path / "d", name, int(g["weights_seed"]), dropout=0.2)
    monkeypatch.setitem(config.PATHS, "deploy", deploy)
    runs = []
    for abort_at in (-1, 3):
        est = WatchPhonePocketNN(model_hash=h, smooth=1, add_mc_samples=True, monte_carlo_samples=25)
        msgs = []
        for f, row32 in enumerate(g["rows"][:6]):
            if f == abort_at:
                assert poke(est._hip_model().handle, 0, 1) == 0
            msgs.append(np.asarray(est.process_row(array("f", row32.tolist()))))
        assert est._hip_model().last_kernel() == "ape_lstm_mc_small"
        runs.append(msgs)
    for f, (a, b) in enumerate(zip(*runs)):
        assert np.abs(a - b).max() < (5e-5 if f == 3 else 1e-30), f


def test_recover_replays_only_into_memory_the_mirror_still_owns(norm_stats):
    """the handle's journal keeps raw device pointers of every call since the last check; the Python mirror hides buffer
    lifetimes, so a device-output call whose tensors were dropped could be replayed into memory the caching allocator had handed to
    somebody else.  The mirror now holds references to the buffers of journaled calls until the next check / recover: tensors
    allocated behind a dropped call do not alias them, and a recovery leaves them untouched."""
    from wear_mocap_ape_amd import _hip
    name = "pocket"
    m, sd, cfg = make_model(name, 0, norm_stats[name])
    poke = _poke(_hip.lib())
    B, T = 300, cfg["T"]
    x = torch.from_numpy(_synthetic_windows(norm_stats[name], B, T, cfg["I"], 3))
    good = m(x, last_step_only=True, normalize_input=True).numpy().copy()
    xd = x.cuda()
    yd = m(xd, last_step_only=True, normalize_input=True)              # device output: journaled, not checked
    ptr_x, ptr_y = xd.data_ptr(), yd.data_ptr()
    del xd, yd
    # same sizes: without the mirror's references the caching allocator would hand out the very blocks the journal points at
    others = [torch.full((B, T, cfg["I"]), 7.0, device="cuda") for _ in range(4)] + [torch.full((B, 1, cfg["O"]), 7.0, device="cuda") for _ in range(4)]
    assert all(o.data_ptr() not in (ptr_x, ptr_y) for o in others)
    assert poke(m.handle, 0, 1) == 0
    out = m(x, last_step_only=True, normalize_input=True).numpy()       # host output: recover re-issues BOTH journaled calls
    assert np.abs(out - good).max() < 1e-6
    assert m.stats()["reissued_calls"] == 2
    assert all(bool((o == 7.0).all()) for o in others)
    assert m._pending == []


@pytest.mark.parametrize("name,S,n_mc,T,route", [("pocket", 330, 25, 6, "ape_lstm_upper32"), ("watch", 90, 25, 8, "ape_lstm_upper32"),
                                                  ("pocket", 3, 25, 6, "ape_lstm_mc_small"), ("uarm", 1, 50, 6, "ape_lstm_mc_small"),
                                                  ("pocket", 40, 7, 6, "ape_lstm_cluster"), ("uarm", 170, 50, 6, "ape_lstm_upper128"),
                                                  ("uarm", 45, 50, 6, "ape_lstm_upper128"),
                                                  # round 5: the routes' new thresholds -- one-tile clusters (SOLO form, launch A on the
                                                  # one-layer cluster form), the 3 x 128 route from 1025 sample rows, the fused launches below
                                                  ("pocket", 21, 25, 6, "ape_lstm_upper32"), ("watch", 60, 25, 8, "ape_lstm_upper32"),
                                                  ("uarm", 21, 50, 6, "ape_lstm_upper128"), ("uarm", 20, 50, 6, "ape_lstm_cluster")])
def test_bank_routes_under_injected_masks_against_the_oracle(norm_stats, name, S, n_mc, T, route):
    """the bank's weight-stationary Monte-Carlo route (layer 0 once per stream, `ape_mc_expand_kernel`,
    `ape_lstm_upper32`) was pinned to the oracle only through the batch-tile route under the same Philox counters.  The test-hooks
    library lets a bank take the caller's masks: every sample row of `ape_streams_step` against `orc.lstm_forward(..., masks=)`
    (1e-6), for the weight-stationary route, the latency kernel and the first-generation kernel."""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    lib = _hip.lib()
    lib.ape_debug_set_bank_masks.restype, lib.ape_debug_set_bank_masks.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    lib.ape_debug_bank_targets.restype, lib.ape_debug_bank_targets.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    st = norm_stats[name]
    m, sd, cfg = make_model(name, 5, st)
    m.set_body(orc.DEFAULT_BODY)
    assert T == cfg["T"]
    I, O, H, L = cfg["I"], cfg["O"], cfg["H"], cfg["L"]
    rows = S * n_mc
    rng = np.random.default_rng(S * 1000 + n_mc)
    feats = _synthetic_windows(st, S, T + 2, I, 8)
    bank = StreamBank(m, S, T, smooth=1, normalize=True, dtype=torch.float64, monte_carlo_samples=n_mc, dropout=0.2, seed=3)
    hist = [[] for _ in range(S)]
    for f in range(T + 2):
        masks = [(rng.random((rows, T, H)) >= 0.2).astype(np.float32) / np.float32(0.8) for _ in range(L - 1)]
        md = torch.from_numpy(np.stack(masks)).cuda()
        assert lib.ape_debug_set_bank_masks(bank._handle, C.c_void_p(md.data_ptr())) == 0
        bank.push_features(torch.from_numpy(np.ascontiguousarray(feats[:, f])).cuda())
        bank.step()
        assert m.last_kernel() == route
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
    assert lib.ape_debug_set_bank_masks(bank._handle, None) == 0
    m.check()
