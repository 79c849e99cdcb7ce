set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=/tmp/prof_pipe
rocprofv3 --kernel-trace --stats -d $P/trace -- python3 $R/tests/tools/time_mlp.py 262144 > $R/gpurun_out/prof_pipe_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/fetch -- python3 $R/tests/tools/time_mlp.py 262144 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/write -- python3 $R/tests/tools/time_mlp.py 262144 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d $P/mfma -- python3 $R/tests/tools/time_mlp.py 262144 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $P/wave -- python3 $R/tests/tools/time_mlp.py 262144 > /dev/null 2>&1
cd $R
python3 tools/summarize_prof.py r02_mlp_pipe $P/trace $P/fetch $P/write ape_mlp_pipe 65536 262144 --pmc-dir $P/mfma --pmc-dir $P/wave --source csrc/mlp_pipe.hip --lds 148544 --flop-per-launch 7.35513e10 --peak-tflops 157.3 --skip-first 20 --note "DropoutFF 22 -> 256 -> 256 -> 256 -> 14, eval mode, 262 144 rows = 8192 tiles of 32 rows over 128 pairs of workgroups (64 tiles per pair); grid 256 workgroups x 256 threads"
cp profiles/r02_mlp_pipe.md gpurun_out/r02_mlp_pipe.md
cp profiles/traffic_latest.json gpurun_out/traffic_latest.json
