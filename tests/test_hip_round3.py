"""GPU parity tests added in round 3 (all through the C ABI of libape_hip.so):
  * the second-generation cluster kernel of the 3 x 128 upper-arm regressor (lstm_cluster16.hip).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ape_oracle as orc
from tests.test_hip_parity import make_model, _synthetic_windows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as entry
    entry.build()


@pytest.mark.parametrize("B,T", [(1024, 64), (513, 49), (700, 50), (2100, 14), (1000, 51), (600, 67), (1024, 200), (1100, 12)])
def test_uarm_second_generation_cluster_kernel(norm_stats, B, T):
    """lstm_cluster16.hip (eval-mode batches above 512 rows of WatchPhoneUarmNN's 3 x 128 LSTM, watch_phone_uarm_nn.py:13-41, with windows of
    12 steps and more -- round 6: of more than 48 steps where lstm_level16.hip serves the call in one launch, i.e. up to 1024 rows)
    against the float32 oracle (module tolerance 1e-6), the first-generation cluster kernel and the batch-tile kernel (other
    summation orders only); window lengths around the depth of its three-layer software pipeline (fill and drain sections),
    ragged and multi-launch batches, the opt-in plain in-XCD exchange (same bits as the default write-through form) and run-to-run determinism."""
    from wear_mocap_ape_amd import _hip
    name = "uarm"
    st = norm_stats[name]
    model, sd, cfg = make_model(name, 5, st)
    x = _synthetic_windows(st, B, T, cfg["I"], 17)
    xd = torch.from_numpy(x).cuda()
    xn = ((x.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
    model.set_kernel("cluster")
    assert model.kernel_name(B, T) == "ape_lstm_cluster16<128, 3, 64, 2>"
    y2 = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    model.check()
    y2b = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    assert np.array_equal(y2, y2b)
    y_wt = torch.empty((B, cfg["O"]), dtype=torch.float32, device="cuda")
    _hip.check(_hip.lib().ape_lstm_forward(model.handle, C.c_void_p(xd.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT | _hip.FLAG_IN_XCD_PLAIN,
                                           None, 0.0, 0, C.c_void_p(y_wt.data_ptr()), None), "ape_lstm_forward")
    torch.cuda.synchronize()
    model.check()
    assert np.array_equal(y_wt.cpu().numpy(), y2)
    model.set_kernel("cluster_gen1")
    assert "ape_lstm_cluster<" in model.kernel_name(B, T)
    y1 = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    y0 = model.set_kernel("tile16")(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    model.set_kernel("auto")
    y_ref = orc.lstm_forward(sd, xn)[:, -1]
    e_ref, e_gen, e_t16 = float(np.abs(y2 - y_ref).max()), float(np.abs(y2 - y1).max()), float(np.abs(y2 - y0).max())
    print(f"\n[uarm B={B} T={T} cluster16] vs oracle {e_ref:.2e}, vs gen-1 kernel {e_gen:.2e}, vs batch-tile kernel {e_t16:.2e}")
    assert e_ref < 1e-6 and e_gen < 1e-6 and e_t16 < 1e-6


@pytest.mark.parametrize("name,B,T,drop", [("uarm", 1024, 6, False), ("pocket", 512, 8, False), ("pocket", 300, 6, True), ("watch", 64, 8, True),
                                           ("pocket", 77, 3, False), ("uarm", 2100, 5, True)])
def test_first_generation_kernel_forms_xcd_local_clusters(norm_stats, name, B, T, drop):
    """lstm_cluster.hip from four clusters on: clusters formed within block-index classes (one XCD each, verified at run time), slices handed
    over by write-through stores (round 5).  The same bits as the any-placement formation (APE_FLAG_NO_XCD_CLASSES: global tickets) and as
    the class formation with the opt-in plain in-XCD stores (APE_FLAG_IN_XCD_PLAIN); eval mode also against the oracle; repeat launches
    bit-equal (the class tickets and XCD words are self-cleaning)."""
    from wear_mocap_ape_amd import _hip
    st = norm_stats[name]
    model, sd, cfg = make_model(name, 9, st)
    x = _synthetic_windows(st, B, T, cfg["I"], 23)
    xd = torch.from_numpy(x).cuda()
    model.set_kernel("cluster_gen1")
    assert "ape_lstm_cluster<" in model.kernel_name(B, T)
    base = _hip.FLAG_NORMALIZE_INPUT | (_hip.FLAG_DROPOUT_PHILOX if drop else 0)
    outs = []
    for extra in (0, 0, _hip.FLAG_NO_XCD_CLASSES, _hip.FLAG_IN_XCD_PLAIN):
        y = torch.empty((B, cfg["O"]), dtype=torch.float32, device="cuda")
        _hip.check(_hip.lib().ape_lstm_forward(model.handle, C.c_void_p(xd.data_ptr()), B, T, base | extra, None, 0.2 if drop else 0.0, 11,
                                               C.c_void_p(y.data_ptr()), None), "ape_lstm_forward")
        torch.cuda.synchronize()
        model.check()
        outs.append(y.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]) and np.array_equal(outs[0], outs[3])
    if not drop:
        xn = ((x.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
        assert float(np.abs(outs[0] - orc.lstm_forward(sd, xn)[:, -1]).max()) < 1e-6
    model.set_kernel("auto")


@pytest.mark.parametrize("B,T", [(1024, 6), (513, 6), (600, 1), (700, 2), (1000, 3), (1024, 7), (1024, 8), (993, 11), (777, 48),
                                 (512, 6), (5, 6), (16, 3), (300, 1), (129, 70), (512, 64)])
def test_uarm_level_synchronous_kernel_for_short_windows(norm_stats, B, T):
    """lstm_level16.hip (round 6: WatchPhoneUarmNN's 3 x 128 LSTM at its DEPLOYED window of 6 steps, watch_phone_uarm_nn.py:13-41,107-121 --
    eval-mode calls that fit ONE launch: 513 .. 1024 rows with windows of up to 48 steps on two row tiles per cluster = two agents per workgroup,
    5 .. 512 rows at every window length on one row tile per cluster) against the float32 oracle (module tolerance 1e-6), the
    first-generation cluster kernel and the batch-tile kernel (other summation orders only): windows shorter than the model is deep
    (T = 1, 2: levels where layers are still missing), long windows on the one-tile form, ragged batches (a last cluster with rows past the batch, whole row tiles and clusters that own none), the raw-window route with the float64 z-score, and run-to-run determinism."""
    name = "uarm"
    st = norm_stats[name]
    model, sd, cfg = make_model(name, 5, st)
    x = _synthetic_windows(st, B, T, cfg["I"], 23)
    xd = torch.from_numpy(x).cuda()
    xn = ((x.astype(np.float64) - st["xx_m"]) / st["xx_s"]).astype(np.float32)
    model.set_kernel("auto")
    assert model.kernel_name(B, T) == "ape_lstm_level16<128, 3, 64>"
    y2 = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    assert model.last_kernel() == "ape_lstm_level16"
    model.check()
    y2b = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    assert np.array_equal(y2, y2b)
    # the same windows already normalised (no z-score in the kernel)
    y2n = model(torch.from_numpy(xn).cuda(), last_step_only=True, normalize_input=False).cpu().numpy()[:, 0]
    model.check()
    assert np.array_equal(y2n, y2)
    model.set_kernel("cluster_gen1")
    assert "ape_lstm_cluster<" in model.kernel_name(B, T)
    y1 = model(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    y0 = model.set_kernel("tile16")(xd, last_step_only=True, normalize_input=True).cpu().numpy()[:, 0]
    model.set_kernel("auto")
    y_ref = orc.lstm_forward(sd, xn)[:, -1]
    e_ref, e_gen, e_t16 = float(np.abs(y2 - y_ref).max()), float(np.abs(y2 - y1).max()), float(np.abs(y2 - y0).max())
    print(f"\n[uarm B={B} T={T} level16] vs oracle {e_ref:.2e}, vs gen-1 kernel {e_gen:.2e}, vs batch-tile kernel {e_t16:.2e}")
    assert e_ref < 1e-6 and e_gen < 1e-6 and e_t16 < 1e-6
