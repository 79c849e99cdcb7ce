# rocprofv3 passes behind profiles/r02_cluster32.md (run on the GPU box: bash tools/prof_cluster32.sh); the trace databases are
# summarised here because they are too big to travel back
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=/tmp/prof_r02f
cd $R
rocprofv3 --kernel-trace --stats -d $P/trace -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/prof_c32_bench.json 2> gpurun_out/prof_c32_trace.log
echo trace done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
echo traffic done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d $P/mfma -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $P/wave -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
echo counters done
python3 tools/summarize_prof.py r02_cluster32 $P/trace $P/fetch $P/write ape_lstm_cluster32 65536 1024 --pmc-dir $P/mfma --pmc-dir $P/wave \
    --source csrc/lstm_cluster32.hip --lds 136208 --flop-per-launch 1.06039345152e11 --peak-tflops 157.3 --skip-first 90 --min-us 600 \
    --note "Command (MI355X, one GPU, final binary of round 2): \`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline\` (40 pre-roll + 50 warm-up + 200 timed steps: launches 91..290 of the 1024 x 64 shape are the timed ones, the f32 leg of \`fp16_config4\` follows); counters from separate \`--kernel-trace --pmc\` passes of \`bench.py --steps 20 --warmup 5 --no-cpu-baseline\` (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32; SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY); recipe \`tools/prof_cluster32.sh\`, summarised on the GPU box by \`tools/summarize_prof.py\`."
cp profiles/r02_cluster32.md gpurun_out/r02_cluster32.md
cp profiles/traffic_latest.json gpurun_out/traffic_latest.json
