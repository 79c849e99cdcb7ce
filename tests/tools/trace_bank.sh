# rocprofv3 kernel trace of a stream bank's frames -> per-kernel medians: bash tests/tools/trace_bank.sh <tag> <S> <n_mc> <frames> <model>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr_$1
rocprofv3 --kernel-trace --stats -d /tmp/tr_$1 -o t -- python3 $R/tests/tools/bank_trace.py $2 $3 $4 auto check $5 > $R/gpurun_out/$1.log 2>&1
python3 $R/tools/trace_stats.py /tmp/tr_$1 3 >> $R/gpurun_out/$1.log 2>&1
grep -v "amdgpu.ids" $R/gpurun_out/$1.log | tail -25
