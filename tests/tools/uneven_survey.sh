# every cooperative kernel beside a queue of device copies on a second stream, many launches each, every launch against the oracle:
#   bash tests/tools/uneven_survey.sh [launches]      (prints one summary line per case; tests/tools/uneven_lstm.py)
N=${1:-1000}
run() { python tests/tools/uneven_lstm.py "$@" 2>&1 | grep -v "^launch\|amdgpu.ids" | tail -1; }
run pocket 640 16 f32 cluster $N 24
run pocket 640 9 f32 cluster $N 24
run pocket 1024 6 f32 cluster $N 8
run watch 600 8 f32 cluster $N 8
run pocket 640 16 f32 cluster $N 24 0x00400000
run pocket 512 8 f32 cluster_gen1 $N 16
run pocket 300 16 f32 cluster_gen1 $N 16
run uarm 1024 6 f32 cluster_gen1 $N 8
run uarm 640 16 f32 cluster $N 24
run uarm 700 13 f32 cluster $N 16
run watch 640 12 f16 cluster $N 24
run watch 700 8 f16 cluster $N 16
run pocket 200 6 f16_gen1 cluster $N 16
run pocket 4 6 f32 auto $N 4
run pocket 1 6 f32 auto $N 4
python tests/tools/uneven_imupose.py 1024 9 $N 8 2>&1 | grep -v "^launch\|amdgpu.ids" | tail -1
python tests/tools/uneven_imupose.py 1500 5 $N 8 2>&1 | grep -v "^launch\|amdgpu.ids" | tail -1
