cd /root/repo
for B in 64 512; do
  APE_LV16_MIN_ROWS=1 APE_LV16_MAX_T=999 python tests/tools/time_uarm.py $B auto 24,48,64 0 2>&1 | grep -v amdgpu.ids
  python tests/tools/time_uarm.py $B cluster_gen1 24,48,64 0 2>&1 | grep -v amdgpu.ids
done
