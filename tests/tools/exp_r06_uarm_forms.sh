set -e
cd /root/repo
for k in cluster_gen1 cluster; do
python tests/tools/time_uarm.py 1024 $k 6,8,12 0
done
echo "--- c16 forced at T>=1"
APE_C16_MIN_T=1 python tests/tools/time_uarm.py 1024 cluster 6,8,12 0
echo "--- c16 ALT"
APE_C16_MIN_T=1 python tests/tools/time_uarm.py 1024 cluster 6,8,12 0x01000000
echo "--- c16 plain"
APE_C16_MIN_T=1 python tests/tools/time_uarm.py 1024 cluster 6,8,12 0x00400000
echo "--- c16 ALT plain"
APE_C16_MIN_T=1 python tests/tools/time_uarm.py 1024 cluster 6,8,12 0x01400000
echo "--- gen1 plain"
python tests/tools/time_uarm.py 1024 cluster_gen1 6,8,12 0x00400000
