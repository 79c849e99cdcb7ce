# every library under arm-pose-estimation_amd/lib/ab/: parity test (-k expression $1) + bank timing (cases $2...)
K=$1; shift
for rep in 1 2; do
for L in arm-pose-estimation_amd/lib/ab/libape_*.so; do
  echo "== $L (pass $rep)"
  if [ $rep = 1 ]; then APE_HIP_LIB=$PWD/$L timeout -k 10 300 python -m pytest tests/test_hip_parity.py -q -m gpu -k "$K" 2>&1 | tail -1; fi
  APE_HIP_LIB=$PWD/$L BRIEF=1 timeout -k 10 200 python tests/tools/time_bank.py "$@" 2>&1 | grep -v amdgpu.ids
done
done
