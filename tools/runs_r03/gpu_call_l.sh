# call L: attention backward with D as the dP product's initial accumulator (one multiply per score); variants of the operand
# prefetch distance and of the static wave priority (build_ab/*: tools/build_variant.sh)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_backward.py tests/test_hip_cfg5.py -q -m gpu > gpurun_out/r03l_tests.log 2>&1; echo "pytest rc=$?" ); tail -6 gpurun_out/r03l_tests.log
for i in 1 2; do
  for v in k new ah3 ah4 prio2 prio0; do
    lib=$PWD/build_ab/$v/gpt_image_edit_amd/libfk_gfx950.so
    [ $v = new ] && lib=$PWD/gpt_image_edit_amd/libfk_gfx950.so
    AB_MODES=1 FK_LIB_PATH=$lib timeout 200 python tools/ab_attention_bwd.py $v >> gpurun_out/r03l_ab_attn_bwd.log 2>&1
  done
done
grep attention_bwd gpurun_out/r03l_ab_attn_bwd.log | sort -k6,6 -k1,1
