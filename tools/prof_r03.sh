# rocprofv3 passes behind profiles/r03_*.md (run on the GPU box: bash tools/prof_r03.sh <what>); the trace databases are summarised
# here because they are too big to travel back.  <what> = cluster32 | bank_mc | f16 | pipe | uarm | imupose
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=/tmp/prof_r03_$1
rm -rf $P
passes() {      # passes <program and arguments...>: trace + the four counter passes of the same command
  rocprofv3 --kernel-trace --stats -d $P/trace -- "$@" > $R/gpurun_out/prof_$WHAT.out 2> $R/gpurun_out/prof_$WHAT.log
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/fetch -- "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/write -- "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d $P/mfma -- "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $P/wave -- "$@" > /dev/null 2>&1
}
WHAT=$1
cd $R
case $1 in
cluster32)
  passes python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline
  python3 tools/summarize_prof.py r03_cluster32 $P/trace $P/fetch $P/write ape_lstm_cluster32 65536 1024 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_cluster32.hip --lds 136208 --flop-per-launch 1.06039345152e11 --peak-tflops 157.3 --skip-first 60 --min-us 600 \
    --note "Command (MI355X, one GPU): \`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline\` (40 pre-roll + 20 warm-up + 100 timed steps of the 1024 x 64 shape; the f32 leg of \`fp16_config4\` follows); counters from separate \`--kernel-trace --pmc\` passes of the same command (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32; SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY); recipe \`tools/prof_r03.sh cluster32\`."
  python3 tools/summarize_prof.py r03_cluster_f16v2_config4 $P/trace $P/fetch $P/write ape_lstm_cluster_f16v2 65536 1024 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_cluster_f16v2.hip --flop-per-launch 1.0576986112e11 --peak-tflops 2500 --skip-first 30 --min-us 150 \
    --note "BASELINE configs[4] (watch-only model, 1024 windows x 64 frames, fp16 W / x / h, fp32 accumulate): the \`fp16_config4\` leg of the same bench.py command as r03_cluster32 (40 pre-roll + 20 timed launches per pass)."
  ;;
bank_mc)
  passes python3 tests/tools/bank_trace.py 1024 25 40
  python3 tools/summarize_prof.py r03_bank_l0 $P/trace $P/fetch $P/write "ape_lstm_upper32<4, true>" 65536 1024 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_upper32.hip --lds 143920 --flop-per-launch 3.498049536e9 --peak-tflops 157.3 --skip-first 6 \
    --note "Launch A of the same frames (see r03_bank_mc.md): layer 0 once per stream, 1024 streams = 32 tiles of 32 on 32 clusters (ONE tile per cluster: every exchange is exposed), T = 6; algorithmic FLOP = 1024 x 6 x 2 x 4H x (I + H) with I = 22.  Latency-bound by construction at this size (3.8 us of MFMAs + one exposed exchange per step); it replaces 124 us of the batch-tile kernel on 64 CUs."
  python3 tools/summarize_prof.py r03_bank_mc $P/trace $P/fetch $P/write "ape_lstm_upper32<32, false>" 65536 25600 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_upper32.hip --lds 143920 --flop-per-launch 1.61244774400e11 --peak-tflops 157.3 --skip-first 6 \
    --note "Command (MI355X, one GPU): \`rocprofv3 --kernel-trace --stats -- python3 tests/tools/bank_trace.py 1024 25 40\` = a stream bank of 1024 streams x 25 Monte-Carlo dropout samples (the deployed estimators' default, watch_phone_pocket_nn.py:13-19), T = 6, 6 + 40 frames; one frame = feature builder, input tiles (\`ape_x_frag_kernel\`), layer 0 once per stream (\`ape_lstm_upper32<4, true>\`, r03_bank_l0.md), input builder (\`ape_mc_expand_kernel\`), THIS kernel over the 25 600 sample rows (800 tiles of 32 rows on 32 clusters of 8 workgroups), head reduce, post kernel.  Algorithmic FLOP of the launch = 25 600 rows x (6 steps x 2 x 4H x (H + H) + 2 O H) = 161.2 GFLOP (the reference runs the repeated window through both layers, nn_models.py:191-207; layer 0 is shared here).  Recipe \`tools/prof_r03.sh bank_mc\`."
  ;;
pipe)
  passes python3 tests/tools/time_mlp.py 262144
  python3 tools/summarize_prof.py r03_mlp_pipe $P/trace $P/fetch $P/write ape_mlp_pipe 65536 262144 --pmc-dir $P/mfma --pmc-dir $P/wave --source csrc/mlp_pipe.hip --lds 148544 --flop-per-launch 7.35513e10 --peak-tflops 157.3 --skip-first 20 --note "DropoutFF 22 -> 256 -> 256 -> 256 -> 14, eval mode, 262 144 rows = 8192 tiles of 32 rows over 128 pairs of workgroups (64 tiles per pair); grid 256 workgroups x 256 threads; \`python3 tests/tools/time_mlp.py 262144\`; recipe \`tools/prof_r03.sh pipe\`."
  ;;
uarm)
  passes python3 tests/tools/time_uarm.py
  python3 tools/summarize_prof.py r03_uarm_T64 $P/trace $P/fetch $P/write "ape_lstm_cluster16<128, 3, 64, 2>" 131072 1024 --wg-threads 512 --pmc-dir $P/mfma --pmc-dir $P/wave --source csrc/lstm_cluster16.hip --lds 92944 --flop-per-launch 4.5502955520e10 --peak-tflops 157.3 --skip-first 10 --min-us 300 --note "WatchPhoneUarmNN's regressor (I = 38, H = 128, L = 3, O = 12; watch_phone_uarm_nn.py:13-41), 1024 windows x 64 frames, eval mode, on the second-generation kernel of that shape (DESIGN.md 4.14; the first generation ran this at 532 us under rocprofv3, 284 MB per launch); \`python3 tests/tools/time_uarm.py\`; recipe \`tools/prof_r03.sh uarm\`."
  ;;
imupose)
  passes python3 tests/tools/time_imupose.py
  python3 tools/summarize_prof.py r03_imupose_cluster $P/trace $P/fetch $P/write "ape_lstm_cluster<256, 2, 256, 2, false>" 65536 512 --pmc-dir $P/mfma --pmc-dir $P/wave --source csrc/lstm_cluster.hip --flop-per-launch 6.8723671040e10 --peak-tflops 157.3 --skip-first 10 --min-us 300 --note "ImuPoseLSTM (nn_models.py:210-249: Linear 22 -> 256 + ReLU, 2 x 256 LSTM with a 256-wide layer-0 input, Linear 256 -> 14), 1024 windows x 64 frames = TWO launches of 512 windows (16 clusters x 16 members x 32 rows) of the first-generation kernel, this round with XCD-local clusters (DESIGN 4.1); algorithmic FLOP of one launch = 512 x (64 x 2 x 4H x (512 + 512) + 2 x 14 x 256); \`python3 tests/tools/time_imupose.py\`; recipe \`tools/prof_r03.sh imupose\`."
  ;;
esac
cp profiles/r03_*.md gpurun_out/ 2>/dev/null || true
cp profiles/traffic_latest.json gpurun_out/traffic_latest.json
