# one command on every library variant under arm-pose-estimation_amd/lib/ab/ (glob $1), one pass: bash tests/tools/sweep_libs.sh 'libape_c32_*' python tests/tools/time_c32_T.py 64 6
G=$1; shift
for L in arm-pose-estimation_amd/lib/ab/$G.so; do
  echo "== $(basename $L)"
  APE_HIP_LIB=$PWD/$L timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids
done
