# the 2 x 2 of commit 839ed65's two changes (store flavour x position of the look), each variant in N fresh processes, asserting build:
#   bash tests/tools/assert_up128.sh [N]        (variants built by tests/tools/build_variant.sh)
N=${1:-6}
for V in r04form plain_qp24 wt_qp20 shipped; do
  L=arm-pose-estimation_amd/lib/ab/libape_up128_$V.so
  echo "== $V"
  for rep in $(seq $N); do
    APE_HIP_LIB=$PWD/$L timeout -k 10 120 python tests/tools/assert_up128.py 1024 50 2>&1 | grep -v amdgpu.ids | cut -c1-230
  done
done
echo done
