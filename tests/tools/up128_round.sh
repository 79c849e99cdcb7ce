# one GPU round on the 3 x 128 bank kernel: parity of the product build, stamps of the diag variants, timing of the product build
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -q -m gpu -k "shared_layer0" 2>&1 | tail -2
for L in arm-pose-estimation_amd/lib/ab/libape_[de]*.so; do
  echo "== $L"
  APE_HIP_LIB=$PWD/$L timeout -k 10 200 python tests/tools/diag_upper128.py 2>&1 | grep -v amdgpu.ids
done
for rep in 1 2; do BRIEF=1 timeout -k 10 200 python tests/tools/time_bank.py uarm_S1024_mc50_T6 2>&1 | grep -v amdgpu.ids; done
