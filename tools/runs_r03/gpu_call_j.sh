# call J: attention backward -- mask-free main loops, paired dK / dV pass plain (mode 1) and interleaved (mode 2)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_backward.py tests/test_hip_cfg5.py -q -m gpu > gpurun_out/r03j_tests.log 2>&1; echo "pytest rc=$?" ); tail -6 gpurun_out/r03j_tests.log
for i in 1 2; do
  AB_MODES=none FK_LIB_PATH=$PWD/build_ab/base/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention_bwd.py base >> gpurun_out/r03j_ab_attn_bwd.log 2>&1
  timeout 200 python tools/ab_attention_bwd.py new >> gpurun_out/r03j_ab_attn_bwd.log 2>&1
done
grep attention_bwd gpurun_out/r03j_ab_attn_bwd.log
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o ab -- env AB_S=8704 python $GRAFT_REPO_ROOT/tools/ab_attention_bwd.py prof > $GRAFT_REPO_ROOT/gpurun_out/r03j_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_ab -name "*results.db" | head -1) gpurun_out/r03j_attn_bwd_kernel_stats.md "tools/ab_attention_bwd.py (S = 8704, 24 heads; all forms interleaved)" > /dev/null 2>&1
grep -i "attention" gpurun_out/r03j_attn_bwd_kernel_stats.md | head -12
