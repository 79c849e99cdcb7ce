# round 6: launch times of the kernels whose flags are raised per member now (compare with the same script on the parent commit)
cd /root/repo
python tests/tools/time_uarm.py 1024 cluster_gen1 6,12,64 0
APE_KERNEL=cluster_gen1 python tests/tools/time_cluster.py pocket 512 6
APE_KERNEL=cluster_gen1 python tests/tools/time_cluster.py pocket 512 64
APE_KERNEL=cluster_gen1 python tests/tools/time_cluster.py watch 60 8
python tests/tools/time_c32_T.py 6 64
python tests/tools/time_c32_T.py f16 64
