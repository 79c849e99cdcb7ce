# every library under arm-pose-estimation_amd/lib/ab/ through a diagnostic tool: bash tests/tools/diag_variants.sh tests/tools/diag_upper128.py [args]
for L in arm-pose-estimation_amd/lib/ab/libape_*.so; do
  echo "== $L"
  APE_HIP_LIB=$PWD/$L timeout -k 10 200 python "$@" 2>&1 | grep -v amdgpu.ids
done
