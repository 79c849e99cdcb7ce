# A/B of the eval bank's frame with the post-filter in the regressor's tail and as a kernel of its own (APE_BANK_POST_APART=1): kernel trace of
# 200 frames of 1024 streams per model.  The tail form exists at commit f3b941f only (measured slower, taken out: profiles/r06_post_in_tail.md);
# check that commit out to run this.  On the GPU box: bash tests/tools/exp_r06_post_in_tail.sh
set -e
R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_r06.sh bank_eval
for m in pocket watch uarm; do mv gpurun_out/prof_r06_bank_eval_${m}_kernels.txt gpurun_out/r06_post_in_tail_$m.txt; done
export APE_BANK_POST_APART=1
bash tools/prof_r06.sh bank_eval
for m in pocket watch uarm; do mv gpurun_out/prof_r06_bank_eval_${m}_kernels.txt gpurun_out/r06_post_apart_$m.txt; done
head -8 gpurun_out/r06_post_apart_*.txt gpurun_out/r06_post_in_tail_*.txt
