# round 6: the one-agent form of ape_lstm_level16 (one row tile per cluster) against the first generation, 5 .. 512 rows, T = 6 and 12
cd /root/repo
for B in 5 16 64 128 256 384 512; do
  APE_LV16_MIN_ROWS=1 python tests/tools/time_uarm.py $B auto 6,12 0 2>&1 | grep -v amdgpu.ids
  python tests/tools/time_uarm.py $B cluster_gen1 6,12 0 2>&1 | grep -v amdgpu.ids
done
