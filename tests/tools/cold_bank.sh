# N cold starts (fresh process each) of both weight-stationary Monte-Carlo bank routes and of the one-stream latency kernel's bank:
# bash tests/tools/cold_bank.sh [N]
N=${1:-15}
for cfg in "uarm 170 50" "pocket 170 25" "watch 200 25" "pocket 1 25" "uarm 3 50"; do
  off=0
  for rep in $(seq $N); do
    L=$(timeout -k 10 120 python tests/tools/cold_bank.py $cfg 2>&1 | grep -v amdgpu.ids | tail -1)
    case "$L" in *OFF*|*Error*|*error*) off=$((off+1)); echo "$L";; esac
  done
  echo "$cfg: $off of $N cold starts off; last: $L"
done
