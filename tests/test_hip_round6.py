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
