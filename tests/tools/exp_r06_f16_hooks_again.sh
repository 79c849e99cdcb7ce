cd $GRAFT_REPO_ROOT
for v in "" f16look1 f16look2 f16raise1 ""; do
  if [ -n "$v" ]; then export APE_HIP_LIB=$GRAFT_REPO_ROOT/arm-pose-estimation_amd/lib/ab/libape_$v.so; else unset APE_HIP_LIB; fi
  echo "variant: ${v:-shipped}"
  python tests/tools/time_f16.py watch 1024 64 2>&1 | grep "gen 2 (in-L2"
done
