# rocprofv3 passes behind profiles/r06_*.md (run on the GPU box: bash tools/prof_r06.sh <what>); the trace databases are summarised
# here because they are too big to travel back.  <what> = cluster32 | uarm | bank_uarm       (round 5's recipes: tools/prof_r05.sh)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=/tmp/prof_r06_$1
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
  python3 tools/summarize_prof.py r06_cluster32 $P/trace $P/fetch $P/write "ape_lstm_cluster32<256, 2, 32, false>" 65536 1024 --model pocket --T 64 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_cluster32.hip --lds 137232 --flop-per-launch 1.06039345152e11 --peak-tflops 157.3 --skip-first 60 --count 100 --min-us 600 \
    --note "Command (MI355X, one GPU): \`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline\` (40 pre-roll + 20 warm-up + 100 TIMED steps of the 1024 x 64 shape = launches 61..160, the ones summarised here; the f32 leg of \`fp16_config4\` and the \`beside_memory_bound_neighbour\` launches follow, the latter beside device copies); counters from separate \`--kernel-trace --pmc\` passes of the same command (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32; SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY); recipe \`tools/prof_r06.sh cluster32\`.  Round 6 changed one thing in this kernel: the LAST step's flags go up per member (one store instruction behind a barrier)."
  python3 tools/summarize_prof.py r06_cluster32_T6 $P/trace $P/fetch $P/write "ape_lstm_cluster32<256, 2, 32, true>" 65536 1024 --model pocket --T 6 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_cluster32.hip --lds 137232 --flop-per-launch 9.94784051e9 --peak-tflops 157.3 --skip-first 10 \
    --note "The short-window instantiation (end forms) in the same bench.py command: the 1024-stream eval bank's LSTM launch (\`stream_bank_T6.S1024_mc1\`, T = 6) and the dispatch-boundary legs at 513 rows.  Round 6: the last step's flags per member -- 87.6 -> 83-86 us."
  python3 tools/summarize_prof.py r06_cluster_f16v2_config4 $P/trace $P/fetch $P/write ape_lstm_cluster_f16v2 65536 1024 --model watch --T 64 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_cluster_f16v2.hip --flop-per-launch 1.0576986112e11 --peak-tflops 2500 --skip-first 30 --min-us 150 \
    --note "BASELINE configs[4] (watch-only model, 1024 windows x 64 frames, fp16 W / x / h, fp32 accumulate): the \`fp16_config4\` leg of the same bench.py command as r06_cluster32 (40 pre-roll + 20 timed launches per pass).  Round 6: the flag owed for a set's publish goes up per MEMBER (wave 0, four words in one store, behind the section's barrier which every wave passes with its store drained) -- a quarter of the flag stores, same time (DESIGN.md 4.11)."
  kernels > gpurun_out/prof_r06_cluster32_kernels.txt
  ;;
uarm)
  passes python3 tests/tools/time_uarm.py 1024 auto 6,64
  python3 tools/summarize_prof.py r06_uarm_T6 $P/trace $P/fetch $P/write "ape_lstm_level16" 131072 1024 --wg-threads 512 --model uarm --T 6 --pmc-dir $P/mfma --pmc-dir $P/wave --source csrc/lstm_level16.hip --lds 115536 --flop-per-launch 4.268752896e9 --peak-tflops 157.3 --skip-first 10 --note "WatchPhoneUarmNN's regressor (I = 38, H = 128, L = 3, O = 12; watch_phone_uarm_nn.py:13-41) at its DEPLOYED window: 1024 windows x 6 frames, eval mode, on the level-synchronous kernel (DESIGN.md 4.19; round 5 ran this shape on the first-generation kernel at 60-61 us = 45 %); \`python3 tests/tools/time_uarm.py 1024 auto 6,64\`; recipe \`tools/prof_r06.sh uarm\`.  Exchange: tagged 8-byte granules, 2 x 3 x 16 KB per 32-window cluster and level."
  python3 tools/summarize_prof.py r06_uarm_T64 $P/trace $P/fetch $P/write "ape_lstm_cluster16<128, 3, 64, 2>" 131072 1024 --wg-threads 512 --model uarm --T 64 --pmc-dir $P/mfma --pmc-dir $P/wave --source csrc/lstm_cluster16.hip --lds 95008 --flop-per-launch 4.5502955520e10 --peak-tflops 157.3 --skip-first 10 --min-us 300 --note "The same model at 1024 windows x 64 frames on the second-generation kernel of that shape (DESIGN.md 4.14); round 6: flags raised per member through an arrival counter in LDS (the final gather waited 3-8 us for 64 per-wave flag stores to one cache line: 454 -> 448 us)."
  kernels > gpurun_out/prof_r06_uarm_kernels.txt
  ;;
bank_uarm)
  passes python3 tests/tools/bank_trace.py 1024 50 30 auto check uarm
  python3 tools/summarize_prof.py r06_bank_uarm $P/trace $P/fetch $P/write "ape_lstm_upper128" 65536 51200 --model uarm --T 6 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_upper128.hip --lds 146448 --flop-per-launch 1.6121856e11 --peak-tflops 157.3 --skip-first 6 --min-us 600 \
    --note "Command: \`rocprofv3 --kernel-trace --stats -- python3 tests/tools/bank_trace.py 1024 50 30 auto check uarm\` (see profiles/r05_bank_uarm.md: same kernel source, same recipe; re-taken on round 6's library).  Recipe \`tools/prof_r06.sh bank_uarm\`."
  python3 tools/summarize_prof.py r06_bank_uarm_l0 $P/trace $P/fetch $P/write "ape_lstm_cluster<128, 1, 64, 2, false>" 65536 1024 --model uarm --T 6 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_cluster.hip --lds 30240 --flop-per-launch 1.044381696e9 --peak-tflops 157.3 --skip-first 6 \
    --note "Launch A of the same frames: layer 0 of the 3 x 128 model once per stream on the one-layer form of the first-generation cluster kernel (round 6: flags per member through an LDS arrival counter)."
  kernels > gpurun_out/prof_r06_bank_uarm_kernels.txt
  ;;
bank_eval)    # the eval bank's frame (feature builder + LSTM + post-filter), per model: kernel trace only
  for m in pocket watch uarm; do
    WHAT=bank_eval_$m
    rm -rf $P/trace
    trace python3 tests/tools/bank_trace.py 1024 0 200 auto check $m
    cat gpurun_out/prof_$WHAT.out | grep -v amdgpu.ids > gpurun_out/prof_r06_${WHAT}_kernels.txt
    kernels >> gpurun_out/prof_r06_${WHAT}_kernels.txt
  done
  ;;
esac
cp profiles/r06_*.md gpurun_out/ 2>/dev/null || true
cp profiles/traffic_latest.json gpurun_out/traffic_latest.json
