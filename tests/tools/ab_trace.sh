# kernel trace of one-stream frames on two builds of the library: bash tests/tools/ab_trace.sh <lib A> <lib B>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in "$1" "$2"; do
  for mode in dev host; do
    rm -rf /tmp/abtr; echo "== $L $mode"
    APE_HIP_LIB=$R/$L rocprofv3 --kernel-trace --stats -d /tmp/abtr -- python3 $R/tests/tools/frame_trace.py 25 1 300 $( [ $mode = host ] && echo host ) > /tmp/abtr.out 2>&1
    python3 $R/tools/trace_stats.py /tmp/abtr 20
  done
done
