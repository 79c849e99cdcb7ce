# one parity test on one library variant, verbose tail: bash tests/tools/run_variant_test.sh <lib> <-k expr>
for rep in 1 2; do
APE_HIP_LIB=$PWD/$1 timeout -k 10 300 python -m pytest tests/test_hip_parity.py -q -m gpu -k "$2" 2>&1 | grep -E "assert|Error|passed|failed" | head -8
done
