# same-box A/B of two builds of the library: bash tests/tools/ab_libs.sh <lib A> <lib B> <command...>
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in "$A" "$B"; do echo "== $L (pass $rep)"; APE_HIP_LIB=$PWD/$L timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids; done
done
