# N cold starts (fresh process each) of the second-generation f32 kernel in its in-L2 (plain-store) hand-over form, short and long windows,
# and of the fp16 kernel; prints the lines that are OFF and a summary:  bash tests/tools/cold_stress_c32.sh [N]
N=${1:-30}
for cfg in "pocket 1024 6 f32" "pocket 1024 64 f32" "pocket 96 3 f32" "watch 1024 64 f16" "uarm 1024 6 f32"; do
  off=0
  for rep in $(seq $N); do
    L=$(timeout -k 10 120 python tests/tools/cold_stress.py $cfg 2>&1 | grep -v amdgpu.ids | tail -1)
    case "$L" in *OFF*|*Error*|*error*) off=$((off+1)); echo "$L";; esac
  done
  echo "$cfg: $off of $N cold starts off; last: $L"
done
