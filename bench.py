#!/usr/bin/env python3
"""Benchmark of the arm-pose hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without RANK/WORLD_SIZE in the environment: this
                                                            process starts the N ranks itself -- as children, before
                                                            it touches the GPU -- and relays rank 0's JSON line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the throughput configuration the metric is quoted on; configs[3]
is the same thing on 8 GPUs): per GPU 1024 independent synthetic 50 Hz IMU windows of 64 frames x 22
features through the watch_phone_pocket_lstm estimator's path -- f64 z-score -> 2x256 LSTM + head
-> de-normalise -> 6D->quaternion + forward kinematics -- with the inputs already resident in HBM.
A "step" is one pass of that path over the rank's 1024 windows (one `ape_lstm_forward` launch + one
`ape_fk` launch).  Streams are sharded contiguously over the ranks (weak scaling, no data-path
collective); the only collective is the start-up broadcast of the weight blob (RCCL).

Prints ONE JSON line on rank 0 (see the task contract), with two extra objects:
  roofline      dominant kernel (the LSTM) -- algorithmic FLOP per launch / its mean duration,
                measured live with HIP events on the launch stream, vs the dense f32 MFMA peak
  cpu_baseline  the oracle's reference-equivalent CPU path (torch nn.LSTM + per-row eigh FK) timed
                on this host's cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
for _p in (str(REPO), str(REPO / "arm-pose-estimation_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WINDOWS_PER_GPU = 1024
T_FRAMES = 64
PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBPS = 8000.0             # same guide: HBM3E ~8 TB/s
PEAK_F16_MFMA_TFLOPS = 2500.0      # same table, "Peak BF16/FP16 MFMA": ~2.5 PF dense (never the 2:1-sparsity figure)
POCKET = dict(I=22, H=256, L=2, O=14, layout=0)


WATCH = dict(I=20, H=256, L=2, O=12, layout=1)
DEFAULT_BODY = np.array([[-0.22, 0.0, 0.0, -0.26, 0.0, 0.0, -0.1704612, 0.4309841, -0.00670862]])   # bone_map.py:42-45


def synthetic_state_dict(I, H, L, O, seed):
    """random-init weights of the architecture (there are no trained checkpoints offline): uniform +-1/sqrt(H) like
    torch's default LSTM / Linear init, numpy PCG64 stream, reference state_dict key order"""
    rng = np.random.default_rng(seed)
    bound = 1.0 / np.sqrt(H)
    sd = {}
    for k in range(L):
        sd[f"lstm.weight_ih_l{k}"] = rng.uniform(-bound, bound, size=(4 * H, I if k == 0 else H)).astype(np.float32)
        sd[f"lstm.weight_hh_l{k}"] = rng.uniform(-bound, bound, size=(4 * H, H)).astype(np.float32)
        sd[f"lstm.bias_ih_l{k}"] = rng.uniform(-bound, bound, size=(4 * H,)).astype(np.float32)
        sd[f"lstm.bias_hh_l{k}"] = rng.uniform(-bound, bound, size=(4 * H,)).astype(np.float32)
    sd["output_layer.weight"] = rng.uniform(-bound, bound, size=(O, H)).astype(np.float32)
    sd["output_layer.bias"] = rng.uniform(-bound, bound, size=(O,)).astype(np.float32)
    return sd


def synthetic_windows(stats, lo, hi, T, I):
    """feature f ~ N(xx_m[f], xx_s[f]) per stream; stream s always gets the same data
    whatever the sharding (seeded per stream block), sw_dt fixed at 0.02 s = 50 Hz"""
    out = np.empty((hi - lo, T, I), dtype=np.float32)
    for blk in range(lo // 256, (hi + 255) // 256):
        rng = np.random.default_rng(1_000_003 + blk)
        x = stats["xx_m"] + stats["xx_s"] * rng.normal(size=(256, T, I))
        a, b = max(lo, blk * 256), min(hi, (blk + 1) * 256)
        out[a - lo:b - lo] = x[a - blk * 256:b - blk * 256]
    out[..., 0] = 0.02
    return out


def cpu_baseline(sd, stats, body, layout, x, budget_s=8.0, gpu_y=None, gpu_est=None):
    """SURVEY 8(d) "CPU baseline, same run": the oracle's reference-equivalent CPU path on this host -- torch-CPU
    nn.LSTM + Linear (the reference's third-party arithmetic, nn_models.py:169-174) + float64 FK -- timed for
      config1_B1_T6_mc1   one 50 Hz stream, window 6, one frame per call (estimator.py:145-178 from the feature row on:
                          window/pad/trim, f64 z-score, model, de-normalise, FK, message),
      config1_B1_T6_mc25  the same with the estimators' default 25 Monte-Carlo dropout samples per frame
                          (nn_models.py:191-207: lstm.train() + x.repeat),
      config1_B1_T6_mc60_smooth5  the reference script's own setting (experimental_applications/watch_phone_pocket_lstm.py:22-25),
      config3_B1024_T64   the benchmark shape, one batched forward + FK over 1024 windows,
    each at 1 thread and at the best thread count of a short probe (torch's default of one thread per core is far from
    the best for 2x256 LSTM GEMMs), each with the reference's FK route (one 4x4 `eigh` per quaternion,
    transformations.py:521-545) and with the vectorisable closed form -- so the GPU/CPU ratio is not won on a strawman.
    The headline `value` is config3 / best threads / eigh, i.e. what a user of the reference gets today on this host;
    its outputs are also what the timed GPU outputs are compared with."""
    from oracle import ape_oracle as orc          # the checker: imported by this leg only
    default_threads = torch.get_num_threads()
    ncpu = os.cpu_count() or 1
    T6 = 6
    run_eval = orc.torch_reference_model(sd)
    run_mc = orc.torch_reference_model(sd, train=True)

    def batch_leg(xb, route):
        return lambda: orc.infer_windows(sd, stats, body, layout, xb, route=route, use_torch=True)

    def stream_leg(n_mc, route, smooth=1):
        # the per-frame loop of one estimator from the feature row on (the row -> feature step is the HIP path's
        # ape_parse_rows and is not part of either side's timing here)
        predict = (lambda hist: run_eval(hist[None].astype(np.float32))[:, -1, :]) if n_mc == 1 else \
                  (lambda hist: run_mc(np.repeat(hist[None].astype(np.float32), n_mc, axis=0))[:, -1, :])
        win = orc.WindowOracle(T6, smooth, stats, predict)
        frames = x[0]                               # 64 feature rows of stream 0, replayed as a 50 Hz sequence
        state = {"i": 0}

        def one():
            pred = win.push(frames[state["i"] % frames.shape[0]])
            state["i"] += 1
            est = orc.arm_pose_from_targets(pred, body, layout, route)
            return orc.msg_with_mc_samples(orc.msg_from_est(est, body, layout), est, True)
        return one

    def timed(fn, units, budget, max_calls=1 << 30):
        # one warm-up call (thread pool, allocator), then at least THREE timed repeats and on until the budget is spent
        fn()
        n, t0 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            el = time.perf_counter() - t0
            if n >= 3 and (el >= budget or n >= max_calls):
                return n * units / el, n * units, el

    # thread-count probe on the benchmark shape
    probe = {}
    for n in sorted({8, 16, 32, 64} | ({default_threads} if default_threads <= 64 else set())):
        if n > ncpu:
            continue
        torch.set_num_threads(n)
        batch_leg(x[:64], "closed")()
        t0 = time.perf_counter()
        batch_leg(x[:256], "closed")()
        probe[n] = 256 / (time.perf_counter() - t0)
    best = max(probe, key=probe.get)

    legs = {}
    for threads, tname in ((1, "threads1"), (best, f"threads{best}_best")):
        torch.set_num_threads(threads)
        for route in ("eigh", "closed"):
            xb = x if threads > 1 else x[:256]      # a 1-thread pass over 1024 x 64 windows takes > 1 s: bounded sample
            v, n, el = timed(batch_leg(xb, route), xb.shape[0], 1.0, max_calls=8)
            legs[f"config3_B1024_T64/{tname}/fk_{route}"] = {"windows_per_s": v, "sample_windows": n, "seconds": el}
            for n_mc in (1, 25):
                v, n, el = timed(stream_leg(n_mc, route), 1, 0.8)
                legs[f"config1_B1_T6_mc{n_mc}/{tname}/fk_{route}"] = {"frames_per_s": v, "sample_frames": n, "seconds": el,
                                                                      "sample_windows_per_s": v * n_mc}
            # the reference script's own setting: 60 samples, smooth 5 (watch_phone_pocket_lstm.py:22-25): 300 rows through the
            # post-filter per frame -- with the reference's per-row eigh this is the 54 frames/s of BASELINE.md
            v, n, el = timed(stream_leg(60, route, smooth=5), 1, 1.5)
            legs[f"config1_B1_T6_mc60_smooth5/{tname}/fk_{route}"] = {"frames_per_s": v, "sample_frames": n, "seconds": el,
                                                                      "sample_windows_per_s": v * 60}
    # headline leg: config3, best threads, reference FK route; its outputs are the parity reference of the timed GPU run
    torch.set_num_threads(best)
    done, t0 = 0, time.perf_counter()
    while True:
        y_ref, est_ref = orc.infer_windows(sd, stats, body, layout, x, route="eigh", use_torch=True)
        done += x.shape[0]
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 64 * x.shape[0]:
            break
    torch.set_num_threads(default_threads)
    # the same windows went through the HIP path in the timed region: report the error of what was timed
    parity = None
    if gpu_y is not None:
        dy = float(np.abs(gpu_y - y_ref).max())
        e = gpu_est.astype(np.float64)
        worst_q = 0.0
        for c in (9, 13, 17):              # quaternions: strict, sign-aware only where the reference w ~ 0
            a, b = e[:, c:c + 4], est_ref[:, c:c + 4]
            dp, dm = np.abs(a - b).max(axis=1), np.abs(a + b).max(axis=1)
            worst_q = max(worst_q, float(np.where(np.abs(b[:, 0]) < 1e-4, np.minimum(dp, dm), dp).max()))
        parity = {"windows": int(x.shape[0]), "max_abs_nn_targets": dy, "max_abs_quaternions": worst_q,
                  "max_abs_origins": float(np.abs(e[:, :9] - est_ref[:, :9]).max()),
                  "budget": "1e-4 targets / 5e-5 quaternions and origins at T=64 (SURVEY 8d); est rows stored as f32"}
    return {"parity": parity, "value": done / el, "unit": "windows/s", "cores": best, "kind": "port",
            "sample": f"{done} windows (B={x.shape[0]}, T={x.shape[1]}) in {el:.1f} s: oracle torch-CPU nn.LSTM+Linear "
                      f"+ per-row eigh FK at the best thread count of the probe "
                      f"{{{', '.join(f'{k}: {v:.0f}/s' for k, v in probe.items())}}} on {ncpu} cpus; `legs` = the other "
                      f"SURVEY 8(d) configurations, ~1 s of CPU work each",
            "legs": legs}


def _tail_report(us, what):
    """the tail of a per-frame latency series: p99.9, max, and WHERE the outliers sit (frame indices above 1.5 x the median) --
    a tail made of the first frames behind the warm-up or of isolated single frames reads differently from a periodic one"""
    us = np.asarray(us)
    med = float(np.median(us))
    idx = np.nonzero(us > 1.5 * med)[0]
    return {"p999_us": float(np.percentile(us, 99.9)), "max_us": float(us.max()), "frames": int(us.size),
            "outliers_above_1p5x_median": {"count": int(idx.size), "frame_indices": [int(i) for i in idx[:32]],
                                           "their_us": [round(float(v), 1) for v in us[idx[:32]]],
                                           "note": f"{what}: frames above 1.5 x the median; index 0 is the first timed frame behind 20 "
                                                   "untimed ones; the host synchronises on every frame (a 50 Hz consumer), so a "
                                                   "late host wake-up shows up here as well as a slow launch"}}


def batch1_latency(model, stats, n_frames=1000):
    """configs[1]: batch=1 streaming, T=6 window, one frame per call, HIP-event timed"""
    from wear_mocap_ape_amd import _hip
    x = torch.from_numpy(synthetic_windows(stats, 0, 1, 6, POCKET["I"])).cuda()
    est = torch.empty((1, 21), dtype=torch.float64, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = _hip.lib()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_frames)]
    for i in range(20 + n_frames):
        if i >= 20:
            evs[i - 20][0].record()
        _hip.check(lib.ape_infer(model.handle, C.c_void_p(x.data_ptr()), 1, 6, _hip.FLAG_NORMALIZE_INPUT, None,
                                 C.c_void_p(est.data_ptr()), _hip.F64, stream), "ape_infer")
        if i >= 20:
            evs[i - 20][1].record()
            evs[i - 20][1].synchronize()      # frame-by-frame, like a 50 Hz stream consumer
    us = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
    out = {"workload": "configs[1]: pocket B=1 T=6 streaming, one ape_infer per frame",
           "p50_us": float(np.percentile(us, 50)), "p99_us": float(np.percentile(us, 99)),
           "frames_per_s": float(1e6 / np.mean(us))}
    out.update(_tail_report(us, "ape_infer"))
    # the roofline that binds this configuration is latency, not a throughput peak: T + L - 1 = 7 dependent phases,
    # each one register-resident GEMV pass + ONE hop through the XCD's L2, behind a weight prologue (profiles/r03_small_stamps.md)
    floor = 2.82 + 7 * (0.69 + 0.24) + 0.34
    out["latency_floor_us"] = floor
    out["frac_of_latency_floor"] = floor / out["p50_us"]
    out["latency_floor_note"] = ("kernel floor = prologue 2.82 + 7 phases x (compute 0.69 + in-XCD granule hop 0.24) + head 0.34 us, from the "
                                 "stamps in profiles/r03_small_stamps.md (kernel itself: 11.1 us); p50 is the whole host-synchronised "
                                 "frame, i.e. it also holds one direct launch + completion (~6 us)")
    # ONE launch per frame (the latency kernel carries the post-filter), issued directly: a hipGraph replay of the same frame
    # is slower on this ROCm (graph_replay_* below; a replay costs ~10-16 us of host time per forward, MI355X guide,
    # "graph-replay-floor"), so the latency path launches directly
    out["launch_form"] = "direct launch per frame (graph replay measured slower, see graph_replay_p50_us)"
    # the same frame step captured once into a hipGraph (LSTM + FK kernels, no memset node) and replayed per frame
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sp = C.c_void_p(side.cuda_stream)
            call = lambda: _hip.check(lib.ape_infer(model.handle, C.c_void_p(x.data_ptr()), 1, 6, _hip.FLAG_NORMALIZE_INPUT,
                                                    None, C.c_void_p(est.data_ptr()), _hip.F64, sp), "ape_infer")
            call()
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                call()
        torch.cuda.current_stream().wait_stream(side)
        gus = []
        for i in range(20 + n_frames):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); graph.replay(); b.record(); b.synchronize()
            if i >= 20:
                gus.append(a.elapsed_time(b) * 1e3)
        out["graph_replay_p50_us"] = float(np.percentile(gus, 50))
        out["graph_replay_p99_us"] = float(np.percentile(gus, 99))
    except Exception as exc:            # reported, never fatal for the headline line
        out["graph_replay_error"] = str(exc)[:200]
    # the estimators' default mode (watch_phone_pocket_nn.py:13-19): ONE stream, 25 Monte-Carlo dropout samples per frame,
    # the whole frame step on the device (raw 55-float row in -> window ring -> 25 samples -> FK -> mean pose datagram)
    try:
        from wear_mocap_ape_amd.streams import StreamBank
        rng = np.random.default_rng(4)
        rows = [torch.from_numpy(rng.normal(size=(1, 55)).astype(np.float32)).cuda() for _ in range(4)]
        bank = StreamBank(model, 1, 6, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=25, dropout=0.2)
        mus = []
        for i in range(20 + n_frames):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            bank.push_rows(rows[i % 4], _hip.PARSE_WATCH_PHONE_POCKET)
            bank.step_datagrams()
            b.record(); b.synchronize()
            if i >= 20:
                mus.append(a.elapsed_time(b) * 1e3)
        model.check()
        out["mc25_stream_p50_us"] = float(np.percentile(mus, 50))
        out["mc25_stream_p99_us"] = float(np.percentile(mus, 99))
        out["mc25_note"] = "one stream, 25 dropout samples per frame (the deployed estimators' default), push_rows + step_datagrams per frame"
        del bank
        # the reference script's own setting (experimental_applications/watch_phone_pocket_lstm.py:22-25 + :52): 60 samples per
        # frame, smoothing over the last 5 predictions -> 300 stacked rows per frame
        bank = StreamBank(model, 1, 6, smooth=5, normalize=True, dtype=torch.float32, monte_carlo_samples=60, dropout=0.2)
        mus = []
        for i in range(20 + n_frames):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            bank.push_rows(rows[i % 4], _hip.PARSE_WATCH_PHONE_POCKET)
            bank.step_datagrams()
            b.record(); b.synchronize()
            if i >= 20:
                mus.append(a.elapsed_time(b) * 1e3)
        model.check()
        out["mc60_smooth5_stream_p50_us"] = float(np.percentile(mus, 50))
        out["mc60_smooth5_stream_p99_us"] = float(np.percentile(mus, 99))
        out["mc60_smooth5_frames_per_s"] = float(1e6 / np.mean(mus))
        out["mc60_smooth5_note"] = ("one stream, 60 dropout samples per frame, smooth 5 (the setting of "
                                    "experimental_applications/watch_phone_pocket_lstm.py:22-25): 300 stacked rows per message")
        del bank
    except Exception as exc:
        out["mc25_error"] = str(exc)[:200]
    return out


def estimator_loop(sd, n_frames=1000):
    """The drop-in `Estimator` surface north_star names, driven exactly like the consumer loop of the reference
    (estimator.py:174-177: parse_row_to_xx -> add_xx_to_row_hist_and_make_prediction -> msg_from_pred): one
    `WatchPhonePocketNN.process_row(row)` per raw 55-float message (array('f'), the listener's wire type, imu.py:66-69) over the
    rows of tests/golden/stream_trace_pocket.npz cycled, host in / host out, wall-clock per frame.  `device_frame` is the
    path `processing_loop` runs (one one-stream bank per estimator: 220 B in, feature build / window / regressor / FK / mean on
    the GPU, 25 + 6N values out -- one call into the C ABI); `staged` the reference-shaped methods one by one (host feature
    build and histories, two device round trips), kept for callers that use them singly."""
    import shutil
    import tempfile
    from array import array
    from wear_mocap_ape_amd import config
    from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import WatchPhonePocketNN
    out = {"workload": "configs[1] through the Python call surface: WatchPhonePocketNN.process_row per 55-float message, T = 6"}
    old_deploy = config.PATHS["deploy"]
    tmp = Path(tempfile.mkdtemp(prefix="ape_bench_"))
    try:
        src, dst = Path(old_deploy), tmp / "deploy"
        h = "670b66fa7664252d1cfb3b5a8a362002ffeeba5c"
        shutil.copytree(src / "data_stats", dst / "data_stats")
        (dst / "nn" / h).mkdir(parents=True)
        shutil.copy(src / "nn" / h / "results.json", dst / "nn" / h / "results.json")      # deployed settings: dropout 0.2, T = 6
        torch.save(({k: torch.from_numpy(v) for k, v in sd.items()}, {"state": {}, "param_groups": []}), dst / "nn" / h / "checkpoint.pt")
        config.PATHS["deploy"] = dst
        g = np.load(REPO / "tests" / "golden" / "stream_trace_pocket.npz")
        rows = [array("f", r.tolist()) for r in g["rows"]]
        for mc, smooth in ((1, 1), (25, 1), (60, 5)):
            ent = {}
            for form in ("device_frame", "device_frame_array", "staged"):
                if form == "device_frame_array" and mc == 1:
                    continue                    # (25 values: nothing to save)
                est = WatchPhonePocketNN(model_hash=h, smooth=smooth, add_mc_samples=True, monte_carlo_samples=mc)
                est.use_device_frame = form != "staged"
                est.msg_as_array = form == "device_frame_array"      # opt-in: the message as one float array instead of a list (estimator.py)
                n = n_frames if form != "staged" else max(200, n_frames // 4)
                for i in range(30):
                    msg = est.process_row(rows[i % len(rows)])
                us = np.empty(n)
                t_all = time.perf_counter()
                for i in range(n):
                    t0 = time.perf_counter()
                    msg = est.process_row(rows[i % len(rows)])
                    us[i] = (time.perf_counter() - t0) * 1e6
                t_all = time.perf_counter() - t_all
                ent[form] = {"p50_us": float(np.percentile(us, 50)), "p99_us": float(np.percentile(us, 99)),
                             "frames_per_s": n / t_all, "frames": n, "msg_len": len(msg)}
                if form == "staged":           # where the staged form's time goes (tests/tools/stream_bench.py's split)
                    tt = np.zeros((200, 3))
                    for i in range(200):
                        a = time.perf_counter(); xx = est.parse_row_to_xx(rows[i % len(rows)])
                        b = time.perf_counter(); pred = est.add_xx_to_row_hist_and_make_prediction(xx)
                        c = time.perf_counter(); est.msg_from_pred(pred, True)
                        tt[i] = (b - a, c - b, time.perf_counter() - c)
                    ent[form]["split_p50_us"] = {"parse_row_to_xx": float(np.median(tt[:, 0]) * 1e6),
                                                 "add_xx_to_row_hist_and_make_prediction": float(np.median(tt[:, 1]) * 1e6),
                                                 "msg_from_pred": float(np.median(tt[:, 2]) * 1e6)}
                elif form == "device_frame_array":
                    ent[form]["p99_over_p50"] = ent[form]["p99_us"] / ent[form]["p50_us"]
                else:
                    ent[form]["aborted_checks"] = est._hip_model().stats()["aborted_checks"]
                    # where a frame's wall time goes, frame by frame (ape_streams_frame_stats, ABI 7): inside the C call -- launch calls,
                    # wait for the completion words, output copy -- and what is left for the Python side of process_row; the frames at and
                    # above the wall-clock p99 are looked at on their own: is the tail the device's or the host's?
                    fs = est._frame_runner().frame_stats()
                    k = min(len(fs["wait_us"]), n)
                    inside = fs["launch_us"][-k:] + fs["wait_us"][-k:] + fs["copy_us"][-k:]
                    wall = us[-k:]
                    tail = wall >= np.percentile(wall, 99)
                    pct = lambda a, q: float(np.percentile(a, q))
                    ent[form]["p99_over_p50"] = ent[form]["p99_us"] / ent[form]["p50_us"]
                    ent[form]["fallback_syncs"] = fs["fallback_syncs"]
                    ent[form]["split_us"] = {
                        "launch_calls": {"p50": pct(fs["launch_us"][-k:], 50), "p99": pct(fs["launch_us"][-k:], 99)},
                        "wait_for_completion_words": {"p50": pct(fs["wait_us"][-k:], 50), "p99": pct(fs["wait_us"][-k:], 99)},
                        "copy_out": {"p50": pct(fs["copy_us"][-k:], 50), "p99": pct(fs["copy_us"][-k:], 99)},
                        "python_side": {"p50": pct(wall - inside, 50), "p99": pct(wall - inside, 99)},
                        "frames_at_or_above_wall_p99": {"n": int(tail.sum()), "wall_mean": float(wall[tail].mean()),
                                                        "launch_calls_mean": float(fs["launch_us"][-k:][tail].mean()),
                                                        "wait_mean": float(fs["wait_us"][-k:][tail].mean()),
                                                        "copy_out_mean": float(fs["copy_us"][-k:][tail].mean()),
                                                        "python_side_mean": float((wall - inside)[tail].mean())}}
                del est
            out[f"mc{mc}_smooth{smooth}"] = ent
    except Exception as exc:                # reported, never fatal for the headline line
        out["error"] = str(exc)[:300]
    finally:
        config.PATHS["deploy"] = old_deploy
        shutil.rmtree(tmp, ignore_errors=True)
    return out


UARM = dict(I=38, H=128, L=3, O=12, layout=1)


def _bank_model(cfg, stats_names):
    """a model of one of the deployed shapes with seeded random weights and the deployed statistics (for the other estimators' banks)"""
    from wear_mocap_ape_amd.estimate import nn_models
    from wear_mocap_ape_amd.utility import data_stats
    m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"], device=torch.cuda.current_device())
    m.load_state_dict(synthetic_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], seed=0))
    st = data_stats.get_norm_stats(*stats_names)
    m.set_norm_stats(st["xx_m"], st["xx_s"], st["yy_m"], st["yy_s"])
    m.set_body(DEFAULT_BODY)
    return m


def stream_bank_numbers(model, stats, cases=None):
    """SURVEY 8 rows a1/a15/f1/f2/f4 at scale: S streams stepped together with all state on the device
    (ape_streams_*): raw 55- / 28-float rows in, window rings, regressor, FK, smoothing, packed datagram rows out.  The pocket model in
    eval mode and in the estimators' default Monte-Carlo mode (25 dropout samples per stream, T = 6), and -- round 4 -- the other two
    deployed estimators in THEIR default modes: WatchPhoneUarmNN (3 x 128, 50 samples, T = 6; watch_phone_uarm_nn.py:14-20) and
    WatchOnlyNN (2 x 256, 25 samples, T = 8, smooth 10; watch_only.py:14-21)."""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS
    out = {}
    if cases is None:
        cases = [("S1024_mc1", "pocket", 1024, None, 1, 100), ("S1024_mc25", "pocket", 1024, 25, 1, 30), ("S8192_mc25", "pocket", 8192, 25, 1, 8),
                 ("uarm_S1024_mc50_T6", "uarm", 1024, 50, 1, 12), ("watch_S1024_mc25_T8", "watch", 1024, 25, 10, 20),
                 # small banks (round 5: the weight-stationary routes from 513 / 1025 sample rows on, one-tile clusters)
                 ("S41_mc25", "pocket", 41, 25, 1, 200), ("uarm_S21_mc50_T6", "uarm", 21, 50, 1, 200)]
    shapes = {"pocket": (POCKET, 6, _hip.PARSE_WATCH_PHONE_POCKET, None),
              "uarm": (UARM, 6, _hip.PARSE_WATCH_PHONE_UARM, (NNS_INPUTS.WATCH_PHONE_CAL_ALL, NNS_TARGETS.ORI_CAL_LARM_UARM)),
              "watch": (WATCH, 8, _hip.PARSE_WATCH_ONLY, (NNS_INPUTS.WATCH_ONLY_CAL, NNS_TARGETS.ORI_CAL_LARM_UARM))}
    models = {"pocket": model}
    try:
        rng = np.random.default_rng(3)
        for key, name, S, n_mc, smooth, frames in cases:
            cfg, T, kind, stats_names = shapes[name]
            if name not in models:
                models[name] = _bank_model(cfg, stats_names)
            mdl = models[name]
            H, L, O, I = cfg["H"], cfg["L"], cfg["O"], cfg["I"]
            width = _hip.PARSE_SHAPES[kind][0]
            rows = [torch.from_numpy(rng.normal(size=(S, width)).astype(np.float32)).cuda() for _ in range(4)]
            bank = StreamBank(mdl, S, T, smooth=smooth, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc, dropout=0.2)
            for f in range(T + smooth):
                bank.push_rows(rows[f % 4], kind)
                bank.step_datagrams()
            # un-timed frames until the part's clocks have settled (round 6: a 0.1 ms frame timed right behind the set-up above read 99.5 us
            # over its first 200 frames and 93.0 over the next 200 -- tests/tools/exp_r06_parse_side_stream.py; the headline's 50 warm-up
            # steps are 40 ms of work, these legs had seven frames)
            t_warm = time.perf_counter()
            while time.perf_counter() - t_warm < 0.04:
                for f in range(8):
                    bank.push_rows(rows[f % 4], kind)
                    bank.step_datagrams()
                torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for f in range(frames):
                bank.push_rows(rows[f % 4], kind)
                bank.step_datagrams()
            b.record()
            b.synchronize()
            mdl.check()
            ms = a.elapsed_time(b) / frames
            ent = {"model": f"{name} ({I}, {H}, {L}, {O})", "T": T, "smooth": smooth,
                   "ms_per_frame_of_all_streams": ms, "stream_frames_per_s": S / (ms * 1e-3),
                   "sample_windows_per_s": S * (n_mc or 1) / (ms * 1e-3)}
            # the frame's dominant kernel, bracketed by HIP events on the step's own stream in a SEPARATE pass of the same
            # frames (ape_streams_profile: two event records per launch, kept out of the pass timed above)
            bank.profile(True)
            for f in range(frames):
                bank.push_rows(rows[f % 4], kind)
                bank.step_datagrams()
            kms, launches = bank.profile_read()
            bank.profile(False)
            mdl.check()
            rows_total = S * (n_mc or 1)
            kname = mdl.last_kernel()
            if n_mc:
                # Monte-Carlo bank: layer 0 once per stream, then the layers above over the S x n_mc sample rows -- algorithmic work of
                # that launch = (L - 1) x 2 * 4H * (H + H) FLOP per row and step + the head (the reference repeats the window n_mc
                # times through ALL layers, nn_models.py:191-207; the shared layer 0 is work the bank does not do)
                flop = rows_total * (T * (L - 1) * 2.0 * 4 * H * (H + H) + 2.0 * O * H)
                alg_bytes = S * T * H * 4 + rows_total * O * 4      # layer-0 sequence in, NN targets out
            else:
                flop = mdl.flops_per_window(T) * rows_total
                kname = mdl.kernel_name(rows_total, T)
                alg_bytes = rows_total * (T * I * 4 + O * 4)
            k_ms = kms / frames                  # all launches of the kernel in one frame (8192 x 25: five chunks)
            tf = flop / (k_ms * 1e-3) / 1e12
            traffic, ttag, stale = load_traffic(kname, rows_total, model=name, T=T)
            ent["roofline"] = {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_from": ttag, "traffic_stale": stale,
                               # the memory side beside the matrix cores: counter bytes per launch over this run's launch time, against
                               # the 8 TB/s of HBM3E (the banks' write-through hand-over puts every slice on the memory side)
                               "hbm_counter_GBps": (traffic * (launches / frames) / (k_ms * 1e-3) / 1e9) if traffic else None,
                               "hbm_counter_frac_of_peak": (traffic * (launches / frames) / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS) if traffic else None,
                               "kernel": kname, "kernel_ms": k_ms, "launches_per_frame": launches / frames,
                               "flop_per_frame": flop, "hbm_algorithmic_bytes_per_frame": alg_bytes,
                               "kernel_share_of_frame": k_ms / ms}
            if n_mc:
                # stated plainly: h_{-1} = 0, so step 0's recurrent span is algorithmic work the kernels do not execute:
                # per layer above layer 0, (2 T - 1) / 2 T of the FLOP above run on the matrix cores
                ex = rows_total * ((L - 1) * (T * 2 - 1) * 2.0 * 4 * H * H + 2.0 * O * H)
                ent["roofline"]["flop_executed_per_frame"] = ex
                ent["roofline"]["frac_executed"] = ex / (k_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS
            out[key] = ent
            del bank
    except Exception as exc:                # reported, never fatal for the headline line
        out["error"] = str(exc)[:200]
    return out


def dispatch_boundaries(n_iter=30):
    """APE_KERNEL_AUTO at its dispatch boundaries, on THIS box: at each threshold of the plan (csrc/ape_api.hip: 512 / 513 eval rows for
    the second-generation kernels, windows of 48 / 49 steps, 4 / 5 and 1024 / 1025 rows for the 3 x 128 model, 4 / 5 rows and 128 / 129 samples for the latency
    kernels, 512 / 513 sample rows for the bank's weight-stationary route, 3 / 4 clusters for the first generation's XCD classes)
    AUTO and every kernel the public switch can force are timed on the same inputs (HIP events, median of n_iter launches after a
    warm-up); `auto_over_best` = AUTO's time over the fastest candidate's -- 1.00 means AUTO picked the fastest there."""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS
    lib = _hip.lib()
    out = {}

    def timed_together(cands):
        """cands: [(tag, fn)] -> {tag: median us}.  The candidates of a case take turns, launch by launch (HIP events, host-synchronised), behind
        un-timed rounds that last until the part's clocks have settled: timed one after the other, the candidate that went first behind the
        set-up read up to 4 % slower than the SAME kernel timed second (round 6)."""
        for _ in range(3):
            for _, fn in cands:
                fn()
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.03:
            for _, fn in cands:
                fn()
            torch.cuda.synchronize()
        us = {tag: [] for tag, _ in cands}
        for it in range(n_iter):
            # (the order rotates: a launch behind the batch-tile candidate, which streams the weights through every L2, starts ~2 us colder)
            for tag, fn in cands[it % len(cands):] + cands[:it % len(cands)]:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); b.synchronize()
                us[tag].append(a.elapsed_time(b) * 1e3)
        return {tag: float(np.median(v)) for tag, v in us.items()}

    try:
        models = {"pocket": _bank_model(POCKET, (NNS_INPUTS.WATCH_PHONE_CAL_HIP, NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS)),
                  "uarm": _bank_model(UARM, (NNS_INPUTS.WATCH_PHONE_CAL_ALL, NNS_TARGETS.ORI_CAL_LARM_UARM))}
        rng = np.random.default_rng(11)

        def forward_case(key, name, B, T, drop, bcast, kernels, extra=0):
            m = models[name]
            cfg = POCKET if name == "pocket" else UARM
            x = torch.from_numpy(rng.normal(size=(1 if bcast else B, T, cfg["I"])).astype(np.float32)).cuda()
            y = torch.empty((B, cfg["O"]), dtype=torch.float32, device="cuda")
            flags = (_hip.FLAG_DROPOUT_PHILOX if drop else 0) | (_hip.FLAG_BROADCAST_X if bcast else 0)
            ent = {"candidates_us": {}, "kernels": {}}
            cands = []
            for k in kernels:
                kk, fl = (k, flags) if isinstance(k, str) else (k[0], flags | k[1])
                tag = kk if isinstance(k, str) else f"{kk}+0x{k[1]:x}"

                def call(kk=kk, fl=fl, tag=tag):
                    m.set_kernel(kk)                    # (a host-side switch on the handle)
                    _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, fl, None, 0.2 if drop else 0.0, 7,
                                                    C.c_void_p(y.data_ptr()), None), "fwd")
                    ent["kernels"][tag] = m.last_kernel()
                cands.append((tag, call))
            ent["candidates_us"] = timed_together(cands)
            m.set_kernel("auto")
            m.check()
            ent["auto_over_best"] = ent["candidates_us"]["auto"] / min(ent["candidates_us"].values())
            out[key] = ent

        # eval rows 512 | 513: first generation | ape_lstm_cluster32
        for B in (512, 513):
            for T in (6, 64):
                forward_case(f"pocket_eval_B{B}_T{T}", "pocket", B, T, False, False, ("auto", "cluster_gen1", "tile16"))
        # 3 x 128, 1024 rows, windows of 48 | 49 steps: ape_lstm_level16 (round 6) | ape_lstm_cluster16; at the deployed 6 steps 4 | 5 rows:
        # latency kernel | ape_lstm_level16 (one row tile per cluster up to 512 rows, two above), 1024 | 1025 rows: one launch of it | two of the
        # first generation
        for T in (48, 49):
            forward_case(f"uarm_eval_B1024_T{T}", "uarm", 1024, T, False, False, ("auto", "cluster_gen1", "tile16"))
        for B in (4, 5, 512, 1024, 1025):
            forward_case(f"uarm_eval_B{B}_T6", "uarm", B, 6, False, False, ("auto", "cluster_gen1", "tile16"))
        # eval rows 4 | 5: latency kernel | first generation
        for B in (4, 5):
            forward_case(f"pocket_eval_B{B}_T6", "pocket", B, 6, False, False, ("auto", "cluster_gen1", "tile16"))
        # one window x 128 | 129 samples: Monte-Carlo latency kernel | first generation
        for B in (128, 129):
            forward_case(f"pocket_mc_one_window_n{B}_T6", "pocket", B, 6, True, True, ("auto", "auto_gen1", "tile16"))
        # first generation, dropout rows in 3 | 4 clusters of 32: any-placement | XCD classes (the selector bit forces the former)
        for B in (96, 128):
            forward_case(f"pocket_dropout_B{B}_T6_clusters{B // 32}", "pocket", B, 6, True, False,
                         ("auto", ("auto", _hip.FLAG_NO_XCD_CLASSES), "tile16"))
        # Monte-Carlo bank, 512 | 513 sample rows: one fused first-generation launch | layer 0 shared + ape_lstm_upper32 (one-tile clusters on
        # the SOLO form: the threshold was 2048 until round 5)
        for S, n_mc in ((32, 16), (27, 19)):
            ent = {"candidates_us": {}, "kernels": {}}
            m = models["pocket"]
            rows = [torch.from_numpy(rng.normal(size=(S, 55)).astype(np.float32)).cuda() for _ in range(4)]
            cands, banks = [], []
            for k in ("auto", "auto_gen1"):
                m.set_kernel(k)                         # (a bank plans its route when it is put into Monte-Carlo mode, and steps under the same switch)
                bank = StreamBank(m, S, 6, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=n_mc, dropout=0.2)
                banks.append(bank)

                def frame(k=k, bank=bank, st={"i": 0}):
                    m.set_kernel(k)
                    bank.push_rows(rows[st["i"] % 4], _hip.PARSE_WATCH_PHONE_POCKET)
                    bank.step_datagrams()
                    st["i"] += 1
                    ent["kernels"][k] = m.last_kernel()
                cands.append((k, frame))
            ent["candidates_us"] = timed_together(cands)
            m.check()
            del banks, cands
            m.set_kernel("auto")
            ent["auto_over_best"] = ent["candidates_us"]["auto"] / min(ent["candidates_us"].values())
            out[f"pocket_mc_bank_{S * n_mc}_sample_rows_T6"] = ent
        out["worst_auto_over_best"] = max(v["auto_over_best"] for v in out.values() if isinstance(v, dict))
    except Exception as exc:                # reported, never fatal for the headline line
        out["error"] = str(exc)[:300]
    return out


def beside_neighbour(model, x, n_launch=40):
    """what the write-through hand-over's bytes cost when the memory side is busy.  Two weight-stationary kernels --
    `ape_lstm_cluster32` on the headline shape (306 MB of counter traffic per launch for 5.8 MB of algorithmic bytes) and the upper-arm bank's
    `ape_lstm_upper128` (1.03 GB per launch) -- timed ALONE and BESIDE a queue of 256 MiB device-to-device copies on a second stream (the
    uneven-load tests' streamer), and the copies' achieved rate alone and beside each kernel.  One measurement each, HIP events."""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.streams import StreamBank
    from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS
    out = {}
    try:
        lib = _hip.lib()
        side = torch.cuda.Stream()
        src = torch.full((64 << 20,), 1.0, dtype=torch.float32, device="cuda")
        dst = torch.empty_like(src)
        copy_bytes = 2.0 * src.numel() * 4              # read + write

        def copies(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(side):
                a.record(side)
                for _ in range(n):
                    dst.copy_(src, non_blocking=True)
                b.record(side)
            return a, b

        a, b = copies(20)
        b.synchronize()
        out["copies_alone_GBps"] = 20 * copy_bytes / (a.elapsed_time(b) * 1e-3) / 1e9

        def beside(step, n, copies_per_burst):
            """`step()` n times alone, then n times while a burst of copies runs; per-launch microseconds both ways and the burst's rate"""
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2 * n)]
            for i in range(n):
                ev[i][0].record(); step(); ev[i][1].record()
            torch.cuda.synchronize()
            ca, cb = copies(copies_per_burst)
            for i in range(n, 2 * n):
                ev[i][0].record(); step(); ev[i][1].record()
            torch.cuda.synchronize()
            alone = float(np.median([p.elapsed_time(q) for p, q in ev[:n]])) * 1e3
            # only the launches that ended before the burst did ran beside it
            burst_ms = ca.elapsed_time(cb)
            t_in = [ca.elapsed_time(q) for p, q in ev[n:]]
            inside = [p.elapsed_time(q) * 1e3 for (p, q), t in zip(ev[n:], t_in) if t < burst_ms]
            return alone, (float(np.median(inside)) if inside else None), len(inside), copies_per_burst * copy_bytes / (burst_ms * 1e-3) / 1e9

        B, T = x.shape[0], x.shape[1]
        y = torch.empty((B, POCKET["O"]), dtype=torch.float32, device=x.device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        step = lambda: _hip.check(lib.ape_lstm_forward(model.handle, C.c_void_p(x.data_ptr()), B, T, _hip.FLAG_NORMALIZE_INPUT, None, 0.0, 0,
                                                       C.c_void_p(y.data_ptr()), st), "ape_lstm_forward")
        al, be, n_in, rate = beside(step, n_launch, 450)
        model.check()
        out["ape_lstm_cluster32_1024x64"] = {"us_alone": al, "us_beside_copies": be, "launches_beside": n_in, "slowdown": (be / al) if be else None,
                                             "copies_GBps_beside": rate}
        um = _bank_model(UARM, (NNS_INPUTS.WATCH_PHONE_CAL_ALL, NNS_TARGETS.ORI_CAL_LARM_UARM))
        width = _hip.PARSE_SHAPES[_hip.PARSE_WATCH_PHONE_UARM][0]
        rows = torch.from_numpy(np.random.default_rng(5).normal(size=(1024, width)).astype(np.float32)).cuda()
        bank = StreamBank(um, 1024, 6, smooth=1, normalize=True, dtype=torch.float32, monte_carlo_samples=50, dropout=0.2)

        def frame():
            bank.push_rows(rows, _hip.PARSE_WATCH_PHONE_UARM)
            bank.step_datagrams()
        for _ in range(8):
            frame()
        al, be, n_in, rate = beside(frame, 12, 240)
        um.check()
        out["uarm_bank_frame_S1024_mc50"] = {"us_alone": al, "us_beside_copies": be, "frames_beside": n_in, "slowdown": (be / al) if be else None,
                                             "copies_GBps_beside": rate, "dominant_kernel": "ape_lstm_upper128 (1.03 GB of counter traffic per launch)"}
        out["note"] = ("copies: 256 MiB device-to-device on a second stream, rate = bytes read + written over the burst's elapsed time; "
                       "`beside` = median over the launches that ended inside the burst")
    except Exception as exc:                # reported, never fatal for the headline line
        out["error"] = str(exc)[:300]
    return out


def other_paths():
    """the SURVEY 8 'next' rows beside the headline, each one measurement (HIP events) on synthetic inputs with seeded random
    weights: the MLP regressor and ImuPoseLSTM (f3) and the ensemble Kalman estimator (f4 tail, parity unpinned)"""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.estimate import nn_models, kalman_models
    out = {}
    rng = np.random.default_rng(9)
    lib = _hip.lib()

    def timed(fn, n_warm, n):
        for _ in range(n_warm):
            fn()
        # ... and on until the part's clocks have settled behind the host-side set-up (as in stream_bank_numbers: 40 ms of launches)
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.04:
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); b.synchronize()
        return a.elapsed_time(b) / n * 1e3          # us

    try:        # DropoutFF 22 -> 256 -> 256 -> 256 -> 14, 262 144 rows (nn_models.py:313-370)
        N = 262144
        m = nn_models.DropoutFF(14, 256, 2, 22, dropout=0.2, device=0)
        m.load_weight_blob(torch.from_numpy(rng.uniform(-0.06, 0.06, m.weight_blob_floats()).astype(np.float32)).cuda())
        x = torch.randn(N, 1, 22, device="cuda"); y = torch.empty(N, 14, device="cuda")
        us = timed(lambda: _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), N, 1, 0, None, 0.0, 0,
                                                           C.c_void_p(y.data_ptr()), None), "fwd"), 10, 50)
        out["mlp_regressor"] = {"rows": N, "us_per_launch": us, "rows_per_s": N / us * 1e6,
                                "tflops": m.flops_per_window(1) * N / us / 1e6, "kernel": m.kernel_name(N, 1),
                                "profile": "profiles/r05_mlp_pipe.md"}
        m.check()
        del m, x, y
    except Exception as exc:
        out["mlp_regressor"] = {"error": str(exc)[:200]}
    try:        # ImuPoseLSTM: Linear(22,256)+ReLU -> 2 x 256 LSTM -> Linear, 1024 windows x 64 frames (nn_models.py:210-249)
        B, T = 1024, 64
        m = nn_models.ImuPoseLSTM(22, 256, 2, 14, device=0)
        m.load_weight_blob(torch.from_numpy(rng.uniform(-0.06, 0.06, m.weight_blob_floats()).astype(np.float32)).cuda())
        x = torch.randn(B, T, 22, device="cuda"); y = torch.empty(B, 14, device="cuda")
        us = timed(lambda: _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, 0, None, 0.0, 0,
                                                           C.c_void_p(y.data_ptr()), None), "fwd"), 5, 20)
        m.check()
        out["imupose_lstm"] = {"windows": B, "frames": T, "us_per_call": us, "windows_per_s": B / us * 1e6,
                               "tflops": m.flops_per_window(T) * B / us / 1e6, "kernel": m.kernel_name(B, T),
                               "profile": "profiles/r05_imupose_split_l0.md, profiles/r05_imupose_split_l1.md"}
        del m, x, y
    except Exception as exc:
        out["imupose_lstm"] = {"error": str(exc)[:200]}
    try:        # the third deployed regressor: WatchPhoneUarmNN's 3 x 128 LSTM (watch_phone_uarm_nn.py:13-41), eval mode
        B = 1024
        m = nn_models.DropoutLSTM(38, 128, 3, 12, device=0)
        m.load_weight_blob(torch.from_numpy(rng.uniform(-0.088, 0.088, m.weight_blob_floats()).astype(np.float32)).cuda())
        res = {}
        for T in (6, 64):
            x = torch.randn(B, T, 38, device="cuda"); y = torch.empty(B, 12, device="cuda")
            us = timed(lambda: _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), B, T, 0, None, 0.0, 0,
                                                               C.c_void_p(y.data_ptr()), None), "fwd"), 10, 50)
            tf = m.flops_per_window(T) * B / us / 1e6
            res[f"T{T}"] = {"us_per_launch": us, "windows_per_s": B / us * 1e6, "tflops": tf, "frac_of_f32_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS,
                            "kernel": m.kernel_name(B, T)}
        m.check()
        out["uarm_lstm"] = dict(res, windows=B, profile="profiles/r05_uarm_T64.md")
        del m, x, y
    except Exception as exc:
        out["uarm_lstm"] = {"error": str(exc)[:200]}
    try:        # KalmanSmartwatchModel.forward + the state shift (kalman_models.py:175-220, watch_phone_pocket_kalman.py:160-162)
        E, W = 48, 10                                   # example_scripts/stream/watch_phone_pocket.py:24-25
        m = kalman_models.KalmanSmartwatchModel(E, W)
        shapes = m.layer_shapes()
        sd = {}
        for name, flip in kalman_models.LAYERS:
            n, k = shapes[name]
            b = 1.0 / np.sqrt(k)
            if flip:
                sd[name + ".mu_weight"] = rng.uniform(-b, b, (n, k)).astype(np.float32)
                sd[name + ".rho_weight"] = np.full((n, k), -3.0, np.float32)
                sd[name + ".mu_bias"] = rng.uniform(-b, b, n).astype(np.float32)
                sd[name + ".rho_bias"] = np.full(n, -3.0, np.float32)
            else:
                sd[name + ".weight"] = rng.uniform(-b, b, (n, k)).astype(np.float32)
                sd[name + ".bias"] = rng.uniform(-b, b, n).astype(np.float32)
        m.load_state_dict(sd)
        res = {}
        for S in (1, 256):
            raw = torch.from_numpy(rng.normal(size=(S, W, 1, 22)).astype(np.float32)).cuda()
            st = [torch.from_numpy((0.1 * rng.normal(size=(S, E, W, 14))).astype(np.float32)).cuda()]

            def frame():
                o = m.forward(raw, st[0])
                st[0] = torch.cat((st[0][:, :, 1:, :], o[0][:, :, None, :]), axis=2)
            us = timed(frame, 10, 100)
            res[f"S{S}"] = {"us_per_frame": us, "stream_frames_per_s": S / us * 1e6}
        m.check()
        out["kalman_estimator"] = dict(res, num_ensemble=E, win_size=W, parity="unpinned (DESIGN 2.1)")
    except Exception as exc:
        out["kalman_estimator"] = {"error": str(exc)[:200]}
    return out


def load_traffic(kernel_name, windows, model=None, T=None):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (profiles/traffic_latest.json, written by tools/summarize_prof.py from separate counter runs of this very
    command; counters cannot be collected from inside the timed run).  Entries are keyed on (kernel instantiation, windows per launch,
    model, T): an entry that names a model / a window length serves only the caller that asks for the same one."""
    tfile = REPO / "profiles" / "traffic_latest.json"
    try:
        tj = json.loads(tfile.read_text())
        best = None
        for ent in tj.get("kernels", [tj]):
            a = ent.get("kernel", "")
            if not (a and (a in kernel_name or kernel_name in a) and ent.get("windows") == windows):
                continue
            if ent.get("model") is not None and model is not None and ent["model"] != model:
                continue
            if ent.get("T") is not None and T is not None and ent["T"] != T:
                continue
            if (model is not None or T is not None) and ent.get("model") is None and ent.get("T") is None:
                continue                                     # (an unkeyed entry of an earlier round answers no keyed question: null, not another shape's bytes)
            best = ent                                       # the newest matching entry wins (the file is append-ordered)
        if best is not None:
            # stale = the entry was measured on another build of the kernel: tools/summarize_prof.py stamps every entry with
            # the SHA-256 of the kernel's object file (lib/build_info.json, written at link time); an entry without a stamp
            # (round 1 / 2) cannot be vouched for either
            stale = True
            try:
                cur = json.loads((REPO / "arm-pose-estimation_amd" / "lib" / "build_info.json").read_text())["objects"]
                stale = not (best.get("object") and cur.get(best["object"]) == best.get("object_sha256"))
            except Exception:
                pass
            return best["hbm_bytes_per_launch"], best.get("tag"), stale
    except Exception:
        pass
    return None, None, None


def fp16_config4(stats_watch, n_iter=20):
    """BASELINE configs[4]: watch-only model, 1024 windows x 64 frames x 20 features, fp16 hidden state /
    weights with fp32 accumulate (ape_model_set_precision F16), HIP-event timed; the exact-f32 kernel on the
    same windows beside it, and the max-abs difference of the NN targets between the two"""
    from wear_mocap_ape_amd import _hip
    from wear_mocap_ape_amd.estimate import nn_models
    cfg = WATCH
    sd = synthetic_state_dict(cfg["I"], cfg["H"], cfg["L"], cfg["O"], seed=0)
    m = nn_models.DropoutLSTM(cfg["I"], cfg["H"], cfg["L"], cfg["O"])
    m.load_state_dict(sd)
    m.set_norm_stats(stats_watch["xx_m"], stats_watch["xx_s"], stats_watch["yy_m"], stats_watch["yy_s"])
    x = torch.from_numpy(synthetic_windows(stats_watch, 0, WINDOWS_PER_GPU, T_FRAMES, cfg["I"])).cuda()
    y = {p: torch.empty((WINDOWS_PER_GPU, cfg["O"]), dtype=torch.float32, device="cuda") for p in ("f32", "f16")}
    lib = _hip.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out, names = {}, {}
    for prec in ("f32", "f16"):
        m.set_precision(prec)
        names[prec] = m.kernel_name(WINDOWS_PER_GPU, T_FRAMES)
        run = lambda: _hip.check(lib.ape_lstm_forward(m.handle, C.c_void_p(x.data_ptr()), WINDOWS_PER_GPU, T_FRAMES,
                                                      _hip.FLAG_NORMALIZE_INPUT, None, 0.0, 0,
                                                      C.c_void_p(y[prec].data_ptr()), stream), "ape_lstm_forward")
        for _ in range(PREROLL):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n_iter):
            run()
        b.record()
        b.synchronize()
        out[prec] = a.elapsed_time(b) / n_iter
    m.check()
    flop = m.flops_per_window(T_FRAMES) * WINDOWS_PER_GPU
    tf16 = flop / (out["f16"] * 1e-3) / 1e12
    traffic, ttag, tstale = load_traffic(names["f16"], WINDOWS_PER_GPU, model="watch", T=T_FRAMES)
    alg_bytes = WINDOWS_PER_GPU * (T_FRAMES * cfg["I"] * 4 + cfg["O"] * 4)
    return {"workload": "configs[4]: watch-only (I=20,H=256,L=2,O=12), 1024 windows x 64 frames, fp16 W/x/h, fp32 accumulate",
            "kernel_ms_f16": out["f16"], "kernel_ms_f32": out["f32"], "windows_per_s_f16": WINDOWS_PER_GPU / out["f16"] * 1e3,
            "algorithmic_tflops_f16": tf16,
            "max_abs_diff_targets_f16_vs_f32": float((y["f16"] - y["f32"]).abs().max().item()),
            "roofline": {"bound": "mfma", "achieved": tf16, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf16 / PEAK_F16_MFMA_TFLOPS, "traffic": traffic, "traffic_from": ttag, "traffic_stale": tstale,
                         "kernel": names["f16"], "kernel_ms": out["f16"], "flop_per_launch": flop,
                         "hbm_algorithmic_bytes_per_launch": alg_bytes,
                         # the bound that binds: 2 (T + 2) = 132 dependent sections, each MFMA spans + gate math + ONE hop of the
                         # exchange (cycles of the round-4 stamps, profiles/r04_f16_duo_ab.md, shipped form: MFMA spans 1132, gates + own
                         # staging 905, flag poll 88 + gather landed 482; at 2.31 GHz).  Round 5: the hand-over is write-through by default
                         # (the guide's valid form): the two row sets of a cluster take turns, so a phase lasts max(2 C, C + E) with C one
                         # section's own work (~1.5 us) and E the exchange from publish to landed (~1.5 us plain in-XCD, ~2.1 us
                         # write-through) -- C + E binds, DESIGN.md 4.11
                         "latency_floor_us": 2 * (T_FRAMES + 2) * (1132 + 905 + 88 + 482) / 2310.0,
                         "frac_of_latency_floor": 2 * (T_FRAMES + 2) * (1132 + 905 + 88 + 482) / 2310.0 / (out["f16"] * 1e3),
                         "hand_over": "write-through (sc1) stores: the default since round 5",
                         "note": "latency-bound, not matrix-bound: per layer-step a wave has 2 x 16 f16 MFMAs (~0.5K cycles) "
                                 "between two cluster-wide exchanges of h; the f16 dense MFMA peak is the stated roofline, "
                                 "latency_floor_us = sections x (MFMA spans + gates + one in-XCD hop) the one that binds"}}


LINE_CAP_BYTES = 6144      # the ONE stdout line stays under this at --gpus 1 (8 KB at --gpus 8); the rest goes to bench_detail.json


def _r(v, nd=4):
    """numbers in the line carry 4-6 significant digits: the full-precision values are in bench_detail.json"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{nd + 2}g}")
    return v


def compact_line(out):
    """The driver's line: the contract keys + roofline + cpu_baseline (headline + the four configs[2]-shaped legs) + parity +
    ONE scalar per secondary leg.  Everything else `main` measured (batch1's tail reports, stream_bank_T6's rooflines,
    dispatch_boundaries, other_paths, the twelve configs[1] CPU legs) stays in `out` and is written to bench_detail.json
    by the caller.  Round 5's line grew to 22.5 KB and the driver could not parse it."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data")}
    line["value"] = _r(line["value"], 6)
    line["ms_per_step"] = _r(line["ms_per_step"], 6)
    cfg = out["config"]
    sh = cfg.get("sharding", {})
    line["config"] = {"workload": cfg["workload"], "windows_per_gpu": cfg["windows_per_gpu"], "frames": cfg["frames"],
                      "features": cfg["features"],
                      "sharding": {"ranks": sh.get("ranks"), "backend": sh.get("backend"),
                                   "collectives": "weight+stats broadcast at start-up; none per step",
                                   "per_rank": [{"rank": p_["rank"], "streams": p_["streams"], "device": p_["device"],
                                                 "kernel_ms": _r(p_["kernel_ms"]), "ms_per_step": _r(p_["ms_per_step"]),
                                                 "bank_ms": [_r(p_.get("bank_S1024_mc25_T6_ms_per_frame")),
                                                             _r(p_.get("bank_uarm_S1024_mc50_T6_ms_per_frame")),
                                                             _r(p_.get("bank_watch_S1024_mc25_T8_smooth10_ms_per_frame"))]}
                                                for p_ in sh.get("per_rank", [])]}}
    if sh.get("ranks", 1) > 1:
        line["config"]["sharding"]["bank_ms_is"] = "per-rank frame of its 1024-stream MC bank: pocket mc25 / uarm mc50 / watch mc25 smooth10"
    rf = out["roofline"]
    line["roofline"] = {k: _r(rf[k], 5) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_from", "traffic_stale",
                                                  "kernel", "kernel_ms", "flop_per_launch", "hbm_algorithmic_bytes_per_launch",
                                                  "hbm_frac") if k in rf}
    bn = rf.get("beside_memory_bound_neighbour") or {}
    if bn and "error" not in bn:
        line["roofline"]["slowdown_beside_copies"] = {"cluster32": _r(g(bn, "ape_lstm_cluster32_1024x64", "slowdown"), 3),
                                                      "uarm_bank_frame": _r(g(bn, "uarm_bank_frame_S1024_mc50", "slowdown"), 3)}
    cs = out.get("cold_start") or {}
    line["cold_start_ms"] = [_r(cs.get("first_step_ms")), _r(cs.get("mean_of_first_10_steps_ms"))]
    cb = out.get("cpu_baseline")
    if cb:
        c = {k: _r(cb[k], 5) for k in ("value", "unit", "cores", "kind")}
        c["sample"] = cb["sample"] if len(cb["sample"]) <= 400 else cb["sample"][:397] + "..."
        c["legs"] = {k: _r(v.get("windows_per_s"), 5) for k, v in cb.get("legs", {}).items() if k.startswith("config3_")}
        line["cpu_baseline"] = c
        pv = out.get("parity_vs_cpu_reference") or {}
        line["parity_vs_cpu_reference"] = {k: (_r(v) if not isinstance(v, str) else v) for k, v in pv.items() if k != "budget"}
        line["parity_vs_cpu_reference"]["budget"] = "1e-4 targets / 5e-5 quats+origins"
        line["gpu_over_cpu"] = _r(out.get("gpu_over_cpu"))
    f16 = out.get("fp16_config4")
    if f16:
        line["fp16_config4"] = {"kernel_ms_f16": _r(f16["kernel_ms_f16"], 5), "kernel_ms_f32": _r(f16["kernel_ms_f32"], 5),
                                "frac": _r(g(f16, "roofline", "frac")), "peak": g(f16, "roofline", "peak"),
                                "kernel": g(f16, "roofline", "kernel"), "max_abs_diff_f16_vs_f32": _r(f16.get("max_abs_diff_targets_f16_vs_f32"))}
    b1 = out.get("batch1")
    if b1:
        el = b1.get("estimator_loop", {})
        line["batch1"] = {"p50_us": _r(b1["p50_us"]), "p99_us": _r(b1["p99_us"]), "frames_per_s": _r(b1["frames_per_s"]),
                          "cpu_frames_per_s": _r(b1.get("cpu_frames_per_s")),
                          "estimator_loop_p50_p99_us": {k: [_r(g(v, "device_frame", "p50_us")), _r(g(v, "device_frame", "p99_us"))]
                                                        for k, v in el.items() if isinstance(v, dict) and "device_frame" in v},
                          "estimator_loop_array_msg_p50_p99_us": {k: [_r(g(v, "device_frame_array", "p50_us")), _r(g(v, "device_frame_array", "p99_us"))]
                                                                  for k, v in el.items() if isinstance(v, dict) and "device_frame_array" in v}}
    sb = out.get("stream_bank_T6")
    if sb:
        line["bank_frame_ms"] = {k: _r(v.get("ms_per_frame_of_all_streams")) for k, v in sb.items() if isinstance(v, dict)}
        line["bank_kernel_share"] = {k: _r(g(v, "roofline", "kernel_share_of_frame"), 3) for k, v in sb.items()
                                     if g(v, "roofline", "kernel_share_of_frame") is not None}
        line["bank_roofline_frac"] = {k: _r(g(v, "roofline", "frac"), 3) for k, v in sb.items() if g(v, "roofline", "frac") is not None}
    op = out.get("other_paths")
    if op:
        line["other_paths_us"] = {"mlp_regressor": _r(g(op, "mlp_regressor", "us_per_launch")),
                                  "imupose_lstm": _r(g(op, "imupose_lstm", "us_per_call")),
                                  "uarm_lstm_T6": _r(g(op, "uarm_lstm", "T6", "us_per_launch")),
                                  "uarm_lstm_T64": _r(g(op, "uarm_lstm", "T64", "us_per_launch")),
                                  "kalman_S256_frame": _r(g(op, "kalman_estimator", "S256", "us_per_frame")),
                                  "kalman_parity": "unpinned"}
    db = out.get("dispatch_boundaries")
    if db:
        line["dispatch_worst_auto_over_best"] = _r(db.get("worst_auto_over_best"))
    if out.get("detail"):
        line["detail"] = out["detail"]
    return json.dumps(line, separators=(",", ":"), allow_nan=False)


def emit(out, json_fd):
    """write the full result to bench_detail.json (repo root and gpurun_out/) and the compact line to the driver's stdout"""
    full = json.dumps(out)
    where = []
    for d in (REPO, REPO / "gpurun_out"):
        try:
            d.mkdir(exist_ok=True)
            (d / "bench_detail.json").write_text(full + "\n")
            where.append(str((d / "bench_detail.json").relative_to(REPO)))
        except OSError:
            pass
    out["detail"] = where[0] if where else "stderr"
    # NOT echoed to stderr: the driver's record keeps one bounded tail of stdout + stderr together, and 20 KB of detail
    # behind the line would push the line out of it
    sys.stderr.write(f"bench.py: full result ({len(full)} bytes) in {', '.join(where) if where else 'no writable place'}\n")
    sys.stderr.flush()
    os.write(json_fd, (compact_line(out) + "\n").encode())


PREROLL = 40        # untimed clock-ramp steps in front of the warmup steps


def visible_gpus_without_opening_them():
    """GPUs this process may use, counted WITHOUT a HIP / HSA call (the spawning parent must provably never open the device):
    the KFD topology in sysfs lists every node with its `simd_count` (0 for a CPU node); HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow it.  None when sysfs has no answer (the children then find out)."""
    n = 0
    if not Path("/dev/kfd").exists():
        return 0                                  # no compute device node: nothing a HIP process could open
    if not Path("/sys/class/kfd").exists():
        return None                               # (a container without the sysfs view: the children will find out)
    try:
        nodes = Path("/sys/class/kfd/kfd/topology/nodes")
        for node in nodes.iterdir():
            for ln in (node / "properties").read_text().splitlines():
                k, _, v = ln.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
    except Exception:
        return None
    if n == 0:
        return None                               # /dev/kfd is there but sysfs lists no compute node: do not guess
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(args):
    """`python bench.py --gpus N` with no RANK/WORLD_SIZE in the environment: start the N single-GPU ranks as CHILD
    processes (torch.distributed.run, one rank per GPU, RCCL) and relay rank 0's JSON line.  This parent never
    touches the GPU (the device count comes from sysfs) and never replaces itself (no exec)."""
    import socket
    import subprocess
    share_gpu = os.environ.get("APE_BENCH_SHARE_GPU") == "1"
    n_dev = visible_gpus_without_opening_them()
    if not share_gpu and n_dev is not None and n_dev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)     # stderr passes through
    line = None
    for ln in proc.stdout.decode("utf-8", "replace").splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        raise SystemExit(f"bench.py: the {args.gpus}-rank launch failed (exit code {proc.returncode}, "
                         f"{'no ' if line is None else ''}result line)")
    os.write(_JSON_FD, (line + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        if args.gpus > 1:
            return spawn_ranks(args)           # before any GPU call in this process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # rehearsal switch for a 1-GPU box: APE_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo (two
    # ranks cannot form an RCCL communicator on one device); the driver's real runs never set it
    share_gpu = os.environ.get("APE_BENCH_SHARE_GPU") == "1"
    # APE_BENCH_FORCE_DIST=1: run the process-group code path (RCCL init, broadcast, barrier, all-reduce) with ONE
    # rank too -- the rehearsal of the N > 1 plumbing that a 1-GPU box allows
    use_dist = world > 1 or os.environ.get("APE_BENCH_FORCE_DIST") == "1"
    dev_index = 0 if share_gpu else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{dev_index} but {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: the process group formed {dist.get_world_size()} ranks, --gpus {args.gpus}")
        if not share_gpu and dist.get_backend() != "nccl":
            raise SystemExit(f"bench.py: the N-rank run must use RCCL (backend nccl), got {dist.get_backend()}")
        world = dist.get_world_size()
    comm_dev = torch.device("cpu") if share_gpu else dev

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if use_dist:
        dist.barrier()
    from wear_mocap_ape_amd import _hip, streams
    from wear_mocap_ape_amd.estimate import nn_models
    from wear_mocap_ape_amd.utility import data_stats
    from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS

    # ---- model: weights exist on rank 0 only and reach the other GPUs by ONE broadcast --------
    model = nn_models.DropoutLSTM(POCKET["I"], POCKET["H"], POCKET["L"], POCKET["O"], device=dev_index)
    n_w = model.weight_blob_floats()
    sd = stats = None
    if rank == 0:
        sd = synthetic_state_dict(POCKET["I"], POCKET["H"], POCKET["L"], POCKET["O"], seed=0)
        stats = data_stats.get_norm_stats(NNS_INPUTS.WATCH_PHONE_CAL_HIP, NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS)
        blob = streams.flatten_state_dict(sd, nn_models.state_dict_keys(POCKET["L"]))
    else:
        blob = None
    blob_dev = streams.broadcast_blob(blob, n_w, dev, always=use_dist)
    stats = streams.broadcast_stats(stats, POCKET["I"], POCKET["O"], dev, always=use_dist)
    model.load_weight_blob(blob_dev)
    model.set_norm_stats(stats["xx_m"], stats["xx_s"], stats["yy_m"], stats["yy_s"])
    model.set_body(DEFAULT_BODY)

    # ---- this rank's shard of the streams, resident in HBM ---------------------------------------
    lo, hi = streams.shard_range(WINDOWS_PER_GPU * world, rank, world)
    x_host = synthetic_windows(stats, lo, hi, T_FRAMES, POCKET["I"])
    x = torch.from_numpy(x_host).to(dev)
    B = hi - lo
    y = torch.empty((B, POCKET["O"]), dtype=torch.float32, device=dev)
    est = torch.empty((B, 21), dtype=torch.float32, device=dev)
    lib = _hip.lib()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    xp, yp, ep = C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(est.data_ptr())
    # what every rank really holds (rank, first stream, end, device, checksum of the weight blob it received)
    mine = torch.tensor([rank, lo, hi, dev_index, float(blob_dev.double().sum().item())], dtype=torch.float64, device=comm_dev)
    shards = [mine]
    if use_dist:
        shards = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(shards, mine)
    shards = [t.cpu().tolist() for t in shards]

    ev_k = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        # exactly what ape_infer enqueues, split so the LSTM launch can be bracketed by HIP events
        if i is not None:
            ev_k[i][0].record()
        _hip.check(lib.ape_lstm_forward(model.handle, xp, B, T_FRAMES, _hip.FLAG_NORMALIZE_INPUT, None, 0.0, 0, yp,
                                        stream), "ape_lstm_forward")
        if i is not None:
            ev_k[i][1].record()
        _hip.check(lib.ape_fk(model.handle, yp, _hip.F32, B, 1, ep, _hip.F32, stream), "ape_fk")

    # The chip raises its clock over the first ~20 ms of continuous work (launch durations in the rocprofv3 trace
    # fall from ~1.03 ms to ~0.90 ms over the first 20 launches after any idle gap): PREROLL untimed steps precede
    # the W warmup steps so that a small --warmup still measures the steady state.  The cold numbers are reported
    # too (`cold_start`): the very first step of the process and the mean of the first ten from an idle chip.
    cold = []
    for i in range(PREROLL + args.warmup):
        if i < 10:
            torch.cuda.synchronize()
            tc = time.perf_counter()
            step()
            torch.cuda.synchronize()
            cold.append((time.perf_counter() - tc) * 1e3)
        else:
            step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    model.check()           # blocking health check of the cluster kernel (bounded spins never expired)

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_k]))
    # BASELINE configs[3] literally: 8192 independent streams, one shard of 1024 per GPU, in the estimators' default Monte-Carlo mode --
    # in an N-rank run every rank also steps ITS 1024-stream x 25-sample bank (device-side frame: rows in, datagram rows out) and
    # reports the frame time, so that the per-node load of configs[3] is in the line rank by rank
    # (round 5: all three deployed estimators' shards -- pocket 25 samples, upper-arm 50 samples, watch-only 25 samples / smooth 10)
    bank_ms = bank_uarm_ms = bank_watch_ms = float("nan")
    if use_dist:
        bk = stream_bank_numbers(model, stats, cases=[("S1024_mc25", "pocket", 1024, 25, 1, 20), ("uarm_S1024_mc50_T6", "uarm", 1024, 50, 1, 10),
                                                      ("watch_S1024_mc25_T8", "watch", 1024, 25, 10, 10)])
        bank_ms = float(bk.get("S1024_mc25", {}).get("ms_per_frame_of_all_streams", float("nan")))
        bank_uarm_ms = float(bk.get("uarm_S1024_mc50_T6", {}).get("ms_per_frame_of_all_streams", float("nan")))
        bank_watch_ms = float(bk.get("watch_S1024_mc25_T8", {}).get("ms_per_frame_of_all_streams", float("nan")))
    # every rank's own numbers travel to rank 0 (a straggler GPU must be visible in the one line the driver gets)
    mine_t = torch.tensor([rank, kernel_ms, own_elapsed / args.steps * 1e3, bank_ms, bank_uarm_ms, bank_watch_ms], dtype=torch.float64, device=comm_dev)
    rank_times = [mine_t]
    if use_dist:
        rank_times = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(rank_times, mine_t)
    rank_times = {int(t_[0].item()): tuple(float(t_[k].item()) for k in range(1, 6)) for t_ in rank_times}
    nn_ = lambda v: None if v != v else v           # (NaN = not measured: a 1-rank run, where stream_bank_T6 carries the frames)
    flop_per_launch = model.flops_per_window(T_FRAMES) * B
    achieved_tf = flop_per_launch / (kernel_ms * 1e-3) / 1e12

    if rank == 0:
        total_windows = sum(int(s_[2] - s_[1]) for s_ in shards) * args.steps
        kname = model.kernel_name(B, T_FRAMES)
        traffic, ttag, tstale = load_traffic(kname, B, model="pocket", T=T_FRAMES)
        out = {
            "metric": "IMU windows/sec", "value": total_windows / elapsed, "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: watch_phone_pocket_lstm path, 1024 windows/GPU x 64 frames x 22 "
                                   "features (I=22,H=256,L=2,O=14), z-score+LSTM+head+denorm+FK, inputs resident in HBM"
                                   + ("" if world == 1 else f"; configs[3] pattern on {world} GPUs: {WINDOWS_PER_GPU * world} streams"),
                       "windows_per_gpu": WINDOWS_PER_GPU, "frames": T_FRAMES, "features": POCKET["I"],
                       "sharding": {"ranks": world, "backend": (dist.get_backend() if use_dist else "none"),
                                    "collectives": "one broadcast of the weight blob + stats at start-up; none per step",
                                    "per_rank": [{"rank": int(s_[0]), "streams": [int(s_[1]), int(s_[2])], "device": int(s_[3]),
                                                  "weight_blob_sum": s_[4], "kernel_ms": rank_times[int(s_[0])][0],
                                                  "ms_per_step": rank_times[int(s_[0])][1],
                                                  "bank_S1024_mc25_T6_ms_per_frame": nn_(rank_times[int(s_[0])][2]),
                                                  "bank_uarm_S1024_mc50_T6_ms_per_frame": nn_(rank_times[int(s_[0])][3]),
                                                  "bank_watch_S1024_mc25_T8_smooth10_ms_per_frame": nn_(rank_times[int(s_[0])][4])} for s_ in shards],
                                    "bank_note": "per rank: one frame of its 1024-stream Monte-Carlo bank for each of the three deployed estimators in "
                                                 "their default modes (configs[3]'s shard: pocket 25 samples, upper-arm 50 samples, watch-only 25 samples "
                                                 "with smooth 10); null in a 1-rank run, where stream_bank_T6 carries them"},
                       "preroll_steps": PREROLL},
            "roofline": {"bound": "mfma", "achieved": achieved_tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_from": ttag,
                         "traffic_stale": tstale, "kernel": kname, "kernel_ms": kernel_ms,
                         "flop_per_launch": flop_per_launch,
                         "hbm_algorithmic_bytes_per_launch": B * (T_FRAMES * POCKET["I"] * 4 + POCKET["O"] * 4),
                         # the other roofline, stated plainly: ~1.8e4 FLOP per algorithmic byte, so HBM is idle by construction
                         "hbm_algorithmic_GBps": B * (T_FRAMES * POCKET["I"] * 4 + POCKET["O"] * 4) / (kernel_ms * 1e-3) / 1e9,
                         "hbm_peak_GBps": 8000.0,
                         "hbm_frac": B * (T_FRAMES * POCKET["I"] * 4 + POCKET["O"] * 4) / (kernel_ms * 1e-3) / 8e12},
            "cold_start": {"first_step_ms": cold[0] if cold else None,
                           "mean_of_first_10_steps_ms": float(np.mean(cold)) if cold else None,
                           "note": "host-synchronised single steps from an idle chip (clock ramp + first-touch), rank 0"},
        }
        if len({round(s_[4], 6) for s_ in shards}) != 1:
            raise SystemExit("bench.py: ranks hold different weight blobs after the broadcast")
        if world == 1:
            out["batch1"] = batch1_latency(model, stats)
            out["batch1"]["estimator_loop"] = estimator_loop(sd)
            out["stream_bank_T6"] = stream_bank_numbers(model, stats)
            out["other_paths"] = other_paths()
            out["dispatch_boundaries"] = dispatch_boundaries()
            out["roofline"]["beside_memory_bound_neighbour"] = beside_neighbour(model, x)
            out["fp16_config4"] = fp16_config4(data_stats.get_norm_stats(NNS_INPUTS.WATCH_ONLY_CAL,
                                                                         NNS_TARGETS.ORI_CAL_LARM_UARM))
            if not args.no_cpu_baseline:
                cb = cpu_baseline(sd, stats, DEFAULT_BODY, POCKET["layout"], x_host, gpu_y=y.cpu().numpy(),
                                  gpu_est=est.cpu().numpy())
                out["parity_vs_cpu_reference"] = cb.pop("parity")
                out["cpu_baseline"] = cb
                out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
                best = [k for k in cb["legs"] if k.startswith("config1_B1_T6_mc1/") and "best" in k and k.endswith("fk_eigh")]
                if best:
                    out["batch1"]["cpu_frames_per_s"] = cb["legs"][best[0]]["frames_per_s"]
                    out["batch1"]["cpu_leg"] = best[0]
                best = [k for k in cb["legs"] if k.startswith("config1_B1_T6_mc60_smooth5/") and "best" in k and k.endswith("fk_eigh")]
                if best:
                    out["batch1"]["mc60_smooth5_cpu_frames_per_s"] = cb["legs"][best[0]]["frames_per_s"]
                    out["batch1"]["mc60_smooth5_cpu_leg"] = best[0]
                # the matching CPU legs beside the drop-in loop (same settings, the oracle's reference-equivalent route)
                for key, leg in (("mc1_smooth1", "config1_B1_T6_mc1/"), ("mc25_smooth1", "config1_B1_T6_mc25/"),
                                 ("mc60_smooth5", "config1_B1_T6_mc60_smooth5/")):
                    best = [k for k in cb["legs"] if k.startswith(leg) and "best" in k and k.endswith("fk_eigh")]
                    if best and key in out["batch1"].get("estimator_loop", {}):
                        out["batch1"]["estimator_loop"][key]["cpu_frames_per_s"] = cb["legs"][best[0]]["frames_per_s"]
                        out["batch1"]["estimator_loop"][key]["cpu_leg"] = best[0]
        emit(out, _JSON_FD)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


# The contract is ONE JSON line on stdout.  Native libraries print there too (RCCL's "Librccl path : ...", gloo's
# connection notes, a stale-library rebuild), so file descriptor 1 is pointed at stderr for the whole run and the
# result line goes out through a private duplicate of the original stdout.
_JSON_FD = 1

if __name__ == "__main__":
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    main()
