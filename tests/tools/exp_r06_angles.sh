# A/B of the half-angle forms (csrc/angle_device.h) in the feature builder and the post-filter: the eval bank's frame per model,
# kernel by kernel, on the library before (lib/ab/libape_before.so) and after.  Run on the GPU box: bash tests/tools/exp_r06_angles.sh
set -e
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "feature or stream_trace or fk or msg or datagram or bank" 2>&1 | tail -3
bash tools/prof_r06.sh bank_eval
for m in pocket watch uarm; do mv gpurun_out/prof_r06_bank_eval_${m}_kernels.txt gpurun_out/r06_angles_after_$m.txt; done
export APE_HIP_LIB=$R/arm-pose-estimation_amd/lib/ab/libape_before.so
bash tools/prof_r06.sh bank_eval
for m in pocket watch uarm; do mv gpurun_out/prof_r06_bank_eval_${m}_kernels.txt gpurun_out/r06_angles_before_$m.txt; done
head -8 gpurun_out/r06_angles_before_*.txt gpurun_out/r06_angles_after_*.txt
