# call K: attention backward without the clamp, softmax scale at the store; then the cfg 5 step
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_backward.py tests/test_hip_cfg5.py -q -m gpu > gpurun_out/r03k_tests.log 2>&1; echo "pytest rc=$?" ); tail -6 gpurun_out/r03k_tests.log
for i in 1 2; do
  AB_MODES=none FK_LIB_PATH=$PWD/build_ab/base/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention_bwd.py base >> gpurun_out/r03k_ab_attn_bwd.log 2>&1
  timeout 200 python tools/ab_attention_bwd.py new >> gpurun_out/r03k_ab_attn_bwd.log 2>&1
done
grep attention_bwd gpurun_out/r03k_ab_attn_bwd.log
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o ab -- env AB_S=8704 python $GRAFT_REPO_ROOT/tools/ab_attention_bwd.py prof > $GRAFT_REPO_ROOT/gpurun_out/r03k_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_ab -name "*results.db" | head -1) gpurun_out/r03k_attn_bwd_kernel_stats.md "tools/ab_attention_bwd.py (S = 8704, 24 heads; all forms interleaved)" > /dev/null 2>&1
grep -i "attention" gpurun_out/r03k_attn_bwd_kernel_stats.md | head -12
( timeout 600 python -m pytest tests/test_hip_train_step.py tests/test_hip_train_seam.py -q -m gpu > gpurun_out/r03k_tests2.log 2>&1; echo "pytest2 rc=$?" ); tail -3 gpurun_out/r03k_tests2.log
( timeout 600 python tools/train_prof.py > gpurun_out/r03k_cfg5.json 2> gpurun_out/r03k_cfg5.err; echo "cfg5 rc=$?" ); tail -2 gpurun_out/r03k_cfg5.err
python -c "
import json; d=json.loads(open('gpurun_out/r03k_cfg5.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','peak_memory_gb','frac_of_mfma_peak_3x_forward','zero2_buckets')})"
