# one parity test on every library under lib/ab, three times each: bash tests/tools/run_variants_test.sh <-k expr>
for L in arm-pose-estimation_amd/lib/ab/libape_*.so; do
echo "== $L"
for rep in 1 2 3; do
APE_HIP_LIB=$PWD/$L timeout -k 10 300 python -m pytest tests/test_hip_parity.py -q -m gpu -k "$1" 2>&1 | grep -E "AssertionError:|passed|failed" | head -3
done
done
