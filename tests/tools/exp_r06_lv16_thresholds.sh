cd /root/repo
for B in 600 700 800 900 1024; do
  python tests/tools/time_uarm.py $B auto 6 0 2>&1 | grep -v amdgpu.ids
  python tests/tools/time_uarm.py $B cluster_gen1 6 0 2>&1 | grep -v amdgpu.ids
done
echo "--- T sweep at 1024: level16 | what AUTO takes without it"
APE_LV16_MAX_T=999 python tests/tools/time_uarm.py 1024 auto 3,24,32,48 0 2>&1 | grep -v amdgpu.ids
APE_LV16_MAX_T=0 python tests/tools/time_uarm.py 1024 auto 3,24,32,48 0 2>&1 | grep -v amdgpu.ids
echo "--- 2048 rows (two launches)"
python tests/tools/time_uarm.py 2048 auto 6 0 2>&1 | grep -v amdgpu.ids
APE_LV16_MAX_T=0 python tests/tools/time_uarm.py 2048 auto 6 0 2>&1 | grep -v amdgpu.ids
