# cold-start repro: every library under lib/ab, N fresh processes of dbg_up128.py 1024 50 2; prints only the frames that are off
N=${1:-6}
for L in arm-pose-estimation_amd/lib/ab/libape_*.so; do
echo "== $L"
for rep in $(seq $N); do
APE_HIP_LIB=$PWD/$L timeout -k 10 200 python tests/tools/dbg_up128.py 1024 50 2 2>&1 | grep -E "rows off|Error|error" | cut -c1-260
done
done
echo done
