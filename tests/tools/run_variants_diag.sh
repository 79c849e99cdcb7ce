# every library under lib/ab: a parity test twice, then the stamps tool
for L in arm-pose-estimation_amd/lib/ab/libape_*.so; do
echo "== $L"
for rep in 1 2 3; do
APE_HIP_LIB=$PWD/$L timeout -k 10 300 python -m pytest tests/test_hip_parity.py -q -m gpu -k "$1" 2>&1 | grep -E "AssertionError:|passed|failed" | head -3
done
APE_HIP_LIB=$PWD/$L timeout -k 10 200 python tests/tools/diag_upper128.py 2>&1 | grep -v amdgpu.ids | head -2
done
