# N cold starts of every flag-based cooperative kernel; prints only the summary and the lines that are OFF
N=${1:-10}
for cfg in "pocket 1024 64 f32" "pocket 1024 6 f32" "pocket 600 6 f32" "uarm 1024 64 f32" "uarm 1024 6 f32" "watch 1024 64 f16" "pocket 1024 64 f16" "uarm 700 12 f32"; do
  off=0
  for rep in $(seq $N); do
    L=$(timeout -k 10 120 python tests/tools/cold_stress.py $cfg 2>&1 | grep -v amdgpu.ids | tail -1)
    case "$L" in *OFF*|*Error*|*error*) off=$((off+1)); echo "$L";; esac
  done
  echo "$cfg: $off of $N cold starts off; last: $L"
done
