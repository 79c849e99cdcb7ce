# The fp16 kernel's hook positions swept again on round 6's shorter section (profiles/r06_f16_hooks_again.log).  Variants first, here:
#   for k in 1 2; do bash tests/tools/build_variant.sh f16look$k lstm_cluster_f16v2.hip -DF16_LOOK_AT=$k; done
# (the raise-position switch of that sweep is gone from the source: 228 us and wrong values beside the counted slab wait); then on the GPU box:
#   bash tests/tools/exp_r06_f16_hooks_again.sh
cd $GRAFT_REPO_ROOT
for v in "" f16look1 f16look2 ""; do
  if [ -n "$v" ]; then export APE_HIP_LIB=$GRAFT_REPO_ROOT/arm-pose-estimation_amd/lib/ab/libape_$v.so; else unset APE_HIP_LIB; fi
  echo "variant: ${v:-shipped}"
  python tests/tools/time_f16.py watch 1024 64 2>&1 | grep "gen 2 (in-L2"
done
