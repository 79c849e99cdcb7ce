for f in 0 1 2 3; do echo "APE_FUSE=$f"; APE_FUSE=$f timeout -k 10 200 python tests/tools/time_mc_small.py pocket 2>&1 | grep -A1 "bank S=1 n_mc=25 smooth=1\|bank S=1 n_mc=60"; done
