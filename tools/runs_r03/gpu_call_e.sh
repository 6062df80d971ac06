cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_train_step.py tests/test_hip_train_seam.py tests/test_hip_cfg5.py tests/test_hip_training.py -q -m gpu -x > gpurun_out/r03e_tests.log 2>&1; echo "pytest rc=$?" ); tail -4 gpurun_out/r03e_tests.log
for i in 1 2; do
  FK_LIB_PATH=$PWD/build_ab/base/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention_bwd.py base >> gpurun_out/r03e_ab_attn_bwd.log 2>&1
  timeout 200 python tools/ab_attention_bwd.py new >> gpurun_out/r03e_ab_attn_bwd.log 2>&1
done
grep attention_bwd gpurun_out/r03e_ab_attn_bwd.log
( timeout 600 python tools/train_prof.py > gpurun_out/r03e_cfg5.json 2> gpurun_out/r03e_cfg5.err; echo "cfg5 rc=$?" ); tail -2 gpurun_out/r03e_cfg5.err
python -c "
import json; d=json.loads(open('gpurun_out/r03e_cfg5.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','peak_memory_gb','frac_of_mfma_peak_3x_forward','zero2_buckets')})"
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o train -- python $GRAFT_REPO_ROOT/tools/train_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r03e_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_t -name "*results.db" | head -1) gpurun_out/r03e_train_kernel_stats.md "cfg 5 train step (1024^2, bs 1, full depth): python tools/train_prof.py = 4 steps (1 warm-up + 3 timed) incl. model / optimiser-state construction" > /dev/null 2>&1
head -34 gpurun_out/r03e_train_kernel_stats.md
