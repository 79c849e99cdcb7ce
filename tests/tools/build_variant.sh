# a variant of the library with one kernel file rebuilt under extra flags: bash tests/tools/build_variant.sh <name> <file.hip> <flags...>
# -> arm-pose-estimation_amd/lib/ab/libape_<name>.so (git-ignored; travels to the GPU box).  DIAG=1: every file with -DAPE_CLUSTER_STAMPS
# (cycle stamps + ape_debug_read_wg), objects cached under lib/ab/diagobj/.  HOOKS=1: + lib/diag/ape_debug.o (the test hooks: injected bank masks, ...).
set -e
N=$1; F=$2; shift 2
cd "$(dirname "$0")/../../arm-pose-estimation_amd/csrc"
mkdir -p ../lib/ab
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC"
if [ -n "$DIAG" ]; then
  mkdir -p ../lib/ab/diagobj
  for o in ../lib/*.o; do
    b=$(basename $o .o)
    [ $b = ${F%.hip} ] && continue
    [ ../lib/ab/diagobj/$b.o -nt $b.hip ] || $CC -DAPE_CLUSTER_STAMPS -c $b.hip -o ../lib/ab/diagobj/$b.o &
  done
  wait
  OBJS=$(ls ../lib/ab/diagobj/*.o | grep -v "/${F%.hip}.o")
  EXTRA=-DAPE_CLUSTER_STAMPS
else
  OBJS=$(ls ../lib/*.o | grep -v "/${F%.hip}.o")
  EXTRA=
fi
$CC $EXTRA -c "$@" $F -o ../lib/ab/${F%.hip}_$N.o
[ -n "$HOOKS" ] && OBJS="$OBJS ../lib/diag/ape_debug.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/libape_$N.so $OBJS ../lib/ab/${F%.hip}_$N.o
rm ../lib/ab/${F%.hip}_$N.o
echo built ../lib/ab/libape_$N.so
