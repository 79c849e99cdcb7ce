# rocprofv3 passes behind profiles/r05_*.md (run on the GPU box: bash tools/prof_r05.sh <what>); the trace databases are summarised
# here because they are too big to travel back.  <what> = cluster32 | bank_mc | bank_uarm | bank_watch | mc_small | pipe | uarm | imupose
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=/tmp/prof_r05_$1
rm -rf $P
trace() {       # trace <program and arguments...>: kernel trace only
  rocprofv3 --kernel-trace --stats -d $P/trace -- "$@" > $R/gpurun_out/prof_$WHAT.out 2> $R/gpurun_out/prof_$WHAT.log
}
passes() {      # passes <program and arguments...>: trace + the four counter passes of the same command
  trace "$@"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/fetch -- "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/write -- "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d $P/mfma -- "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $P/wave -- "$@" > /dev/null 2>&1
}
kernels() {     # per-kernel launch counts and mean durations of the trace pass (the frame's composition)
  python3 - $P/trace <<'PY'
import sqlite3, glob, collections, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    d = collections.defaultdict(list)
    for name, start, end, gx in con.execute("select name, start, end, grid_x from kernels"):
        d[(name[:90], gx)].append((end - start) / 1e3)
    for (name, gx), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if len(v) >= 10: print(f"{len(v):6d} x {sum(v)/len(v):9.1f} us (min {min(v):8.1f})  grid {gx:8d}  {name}")
PY
}
WHAT=$1
cd $R
case $1 in
cluster32)
  passes python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline
  python3 tools/summarize_prof.py r05_cluster32 $P/trace $P/fetch $P/write ape_lstm_cluster32 65536 1024 --model pocket --T 64 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_cluster32.hip --lds 137232 --flop-per-launch 1.06039345152e11 --peak-tflops 157.3 --skip-first 60 --min-us 600 \
    --note "Command (MI355X, one GPU): \`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline\` (40 pre-roll + 20 warm-up + 100 timed steps of the 1024 x 64 shape; the f32 leg of \`fp16_config4\` follows); counters from separate \`--kernel-trace --pmc\` passes of the same command (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32; SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY); recipe \`tools/prof_r05.sh cluster32\`."
  python3 tools/summarize_prof.py r05_cluster_f16v2_config4 $P/trace $P/fetch $P/write ape_lstm_cluster_f16v2 65536 1024 --model watch --T 64 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_cluster_f16v2.hip --flop-per-launch 1.0576986112e11 --peak-tflops 2500 --skip-first 30 --min-us 150 \
    --note "BASELINE configs[4] (watch-only model, 1024 windows x 64 frames, fp16 W / x / h, fp32 accumulate): the \`fp16_config4\` leg of the same bench.py command as r05_cluster32 (40 pre-roll + 20 timed launches per pass)."
  kernels > gpurun_out/prof_cluster32_kernels.txt
  ;;
bank_mc)
  passes python3 tests/tools/bank_trace.py 1024 25 40
  python3 tools/summarize_prof.py r05_bank_l0 $P/trace $P/fetch $P/write "ape_lstm_upper32<4, true>" 65536 1024 --model pocket --T 6 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_upper32.hip --lds 144944 --flop-per-launch 3.498049536e9 --peak-tflops 157.3 --skip-first 6 \
    --note "Launch A of the same frames (see r05_bank_mc.md): layer 0 once per stream, 1024 streams = 32 tiles of 32 on 32 clusters (one tile per cluster: every exchange is exposed), T = 6; algorithmic FLOP = 1024 x 6 x 2 x 4H x (I + H) with I = 22."
  python3 tools/summarize_prof.py r05_bank_mc $P/trace $P/fetch $P/write "ape_lstm_upper32<32, false>" 65536 25600 --model pocket --T 6 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_upper32.hip --lds 144944 --flop-per-launch 1.61244774400e11 --peak-tflops 157.3 --skip-first 6 \
    --note "Command (MI355X, one GPU): \`rocprofv3 --kernel-trace --stats -- python3 tests/tools/bank_trace.py 1024 25 40\` = a stream bank of 1024 streams x 25 Monte-Carlo dropout samples (the pocket estimator's default, watch_phone_pocket_nn.py:13-19), T = 6, 6 + 40 frames.  Algorithmic FLOP of the launch = 25 600 rows x (6 steps x 2 x 4H x (H + H) + 2 O H) = 161.2 GFLOP; executed 11/12 of it (h_{-1} = 0: step 0 is the input span alone).  Recipe \`tools/prof_r05.sh bank_mc\`."
  kernels > gpurun_out/prof_bank_mc_kernels.txt
  ;;
bank_uarm)
  passes python3 tests/tools/bank_trace.py 1024 50 30 auto check uarm
  python3 tools/summarize_prof.py r05_bank_uarm $P/trace $P/fetch $P/write "ape_lstm_upper128" 65536 51200 --model uarm --T 6 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_upper128.hip --lds 146448 --flop-per-launch 1.6121856e11 --peak-tflops 157.3 --skip-first 6 --min-us 600 \
    --note "Command (MI355X, one GPU): \`rocprofv3 --kernel-trace --stats -- python3 tests/tools/bank_trace.py 1024 50 30 auto check uarm\` = a stream bank of 1024 streams x 50 Monte-Carlo dropout samples on the upper-arm model (38 -> 3 x 128 -> 12; the defaults of watch_phone_uarm_nn.py:14-20), T = 6, 6 + 30 frames; one frame = feature builder, layer 0 once per stream, \`ape_mc_expand128_kernel\` (masked layer-0 output in fragment order + layer 1's keep bits), THIS kernel over the 51 200 sample rows (1600 tiles of 32 rows on 64 clusters of 4 workgroups, layers 1 and 2), head reduce, post kernel.  Algorithmic FLOP of the launch = 51 200 rows x (6 steps x 2 layers x 2 x 4H x (H + H) + 2 O H) = 161.2 GFLOP (the reference runs every step through both spans of both layers, nn_models.py:191-207); executed 11/12 of it (h_{-1} = 0: step 0 of a layer is its input span alone).  Recipe \`tools/prof_r05.sh bank_uarm\`."
  python3 tools/summarize_prof.py r05_bank_uarm_l0 $P/trace $P/fetch $P/write "ape_lstm_cluster<128, 1, 64, 2, false>" 65536 1024 --model uarm --T 6 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_cluster.hip --lds 30224 --flop-per-launch 1.044381696e9 --peak-tflops 157.3 --skip-first 6 \
    --note "Launch A of the same frames (see r05_bank_uarm.md): layer 0 of the 3 x 128 model once per stream on the one-layer form of the first-generation cluster kernel, 1024 streams = 32 clusters of 8 workgroups (32 streams each), T = 6, every exchange exposed (one layer has nothing to overlap it with); algorithmic FLOP = 1024 x 6 x 2 x 4H x (I + H) with I = 38, H = 128.  Until round 5 this launch ran on \`ape_lstm_tile16<128,1,4>\` (64 workgroups, 54.5 us)."
  kernels > gpurun_out/prof_bank_uarm_kernels.txt
  ;;
bank_watch)
  passes python3 tests/tools/bank_trace.py 1024 25 30 auto check watch
  python3 tools/summarize_prof.py r05_bank_watch $P/trace $P/fetch $P/write "ape_lstm_upper32<32, false>" 65536 25600 --model watch --T 8 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_upper32.hip --lds 144944 --flop-per-launch 2.149318656e11 --peak-tflops 157.3 --skip-first 6 \
    --note "Command (MI355X, one GPU): \`rocprofv3 --kernel-trace --stats -- python3 tests/tools/bank_trace.py 1024 25 30 auto check watch\` = the watch-only model's bank (2 x 256, T = 8, 25 samples: watch_only.py:14-21), 8 + 30 frames.  Algorithmic FLOP of the launch = 25 600 rows x (8 steps x 2 x 4H x (H + H) + 2 O H); executed 15/16 of it.  Recipe \`tools/prof_r05.sh bank_watch\`."
  kernels > gpurun_out/prof_bank_watch_kernels.txt
  ;;
mc_small)
  # one stream's Monte-Carlo frame: device-side (25 samples, smooth 1) and host in / host out (ape_streams_frame_host)
  trace python3 tests/tools/frame_trace.py 25 1 400
  kernels > gpurun_out/prof_mc_small_device_kernels.txt
  rm -rf $P/trace
  trace python3 tests/tools/frame_trace.py 25 1 400 host
  kernels > gpurun_out/prof_mc_small_host_kernels.txt
  rm -rf $P/trace
  trace python3 tests/tools/frame_trace.py 60 5 400 host
  kernels > gpurun_out/prof_mc_small_host60_kernels.txt
  ;;
pipe)
  passes python3 tests/tools/time_mlp.py 262144
  python3 tools/summarize_prof.py r05_mlp_pipe $P/trace $P/fetch $P/write ape_mlp_pipe 65536 262144 --model ff --T 1 --pmc-dir $P/mfma --pmc-dir $P/wave --source csrc/mlp_pipe.hip --lds 149568 --flop-per-launch 7.35513e10 --peak-tflops 157.3 --skip-first 20 --note "DropoutFF 22 -> 256 -> 256 -> 256 -> 14, eval mode, 262 144 rows = 8192 tiles of 32 rows over 128 pairs of workgroups (64 tiles per pair); grid 256 workgroups x 256 threads; \`python3 tests/tools/time_mlp.py 262144\`; recipe \`tools/prof_r05.sh pipe\`."
  ;;
uarm)
  passes python3 tests/tools/time_uarm.py
  python3 tools/summarize_prof.py r05_uarm_T64 $P/trace $P/fetch $P/write "ape_lstm_cluster16<128, 3, 64, 2>" 131072 1024 --wg-threads 512 --model uarm --T 64 --pmc-dir $P/mfma --pmc-dir $P/wave --source csrc/lstm_cluster16.hip --lds 94992 --flop-per-launch 4.5502955520e10 --peak-tflops 157.3 --skip-first 10 --min-us 300 --note "WatchPhoneUarmNN's regressor (I = 38, H = 128, L = 3, O = 12; watch_phone_uarm_nn.py:13-41), 1024 windows x 64 frames, eval mode, on the second-generation kernel of that shape (DESIGN.md 4.14; the first generation ran this at 532 us under rocprofv3, 284 MB per launch); \`python3 tests/tools/time_uarm.py\`; recipe \`tools/prof_r05.sh uarm\`."
  ;;
imupose)
  passes python3 tests/tools/time_imupose.py
  python3 tools/summarize_prof.py r05_imupose_split_l0 $P/trace $P/fetch $P/write "ape_lstm_upper32<32, true>" 65536 1024 --model imupose --T 64 --pmc-dir $P/mfma --pmc-dir $P/wave --source csrc/lstm_upper32.hip --lds 144944 --flop-per-launch 6.8719476736e10 --peak-tflops 157.3 --skip-first 10 --min-us 300 --note "ImuPoseLSTM (nn_models.py:210-249: Linear 22 -> 256 + ReLU, 2 x 256 LSTM with a 256-wide layer-0 input, Linear 256 -> 14), 1024 windows x 64 frames, round 5: ONE LAYER PER LAUNCH on the persistent 32-row clusters of lstm_upper32.hip -- this is layer 0 in the SEQ form with the wide input (32 tiles on 32 clusters, one tile each: the solo form, own gather under the input span; every step's slices to the sequence layer 1 reads).  Algorithmic FLOP of the launch = 1024 x 64 x 2 x 4H x (256 + 256); \`python3 tests/tools/time_imupose.py\`; recipe \`tools/prof_r05.sh imupose\`.  The first-generation kernel's profile of the same model (two launches of 512 windows, 734.7 us each) is profiles/r05_imupose_cluster.md."
  python3 tools/summarize_prof.py r05_imupose_split_l1 $P/trace $P/fetch $P/write "ape_lstm_upper32<32, false>" 65536 1024 --model imupose --T 64 --pmc-dir $P/mfma --pmc-dir $P/wave --source csrc/lstm_upper32.hip --lds 144944 --flop-per-launch 6.8726816768e10 --peak-tflops 157.3 --skip-first 10 --min-us 300 --note "Layer 1 of the same calls (see r05_imupose_split_l0.md): reads layer 0's sequence as its input tiles, head partials on the last step.  Algorithmic FLOP = 1024 x (64 x 2 x 4H x (256 + 256) + 2 x 14 x 256)."
  ;;
imupose_gen1)
  # the first-generation kernel on the same model (what served it until round 5; still serves it up to 512 windows): forced by set_kernel('cluster')
  passes python3 tests/tools/time_imupose.py 1024 64 cluster
  python3 tools/summarize_prof.py r05_imupose_cluster $P/trace $P/fetch $P/write "ape_lstm_cluster<256, 2, 256, 2, false>" 65536 512 --model imupose --T 64 --pmc-dir $P/mfma --pmc-dir $P/wave --source csrc/lstm_cluster.hip --flop-per-launch 6.8723671040e10 --peak-tflops 157.3 --skip-first 10 --min-us 300 --note "ImuPoseLSTM (nn_models.py:210-249), 1024 windows x 64 frames as TWO launches of 512 windows (16 clusters x 16 members x 32 rows) of the first-generation kernel, forced by \`set_kernel('cluster')\` (AUTO runs this shape on the layer-split route since round 5: r05_imupose_split_l0.md); algorithmic FLOP of one launch = 512 x (64 x 2 x 4H x (512 + 512) + 2 x 14 x 256); \`python3 tests/tools/time_imupose.py 1024 64 cluster\`; recipe \`tools/prof_r05.sh imupose_gen1\`."
  ;;
esac
cp profiles/r05_*.md gpurun_out/ 2>/dev/null || true
cp profiles/traffic_latest.json gpurun_out/traffic_latest.json
