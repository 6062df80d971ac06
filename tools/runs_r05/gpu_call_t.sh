# Round 5, call T: (1) the backward after the dK / dV kernel's restructuring (plain grid by default) against the previous build;
# backward tests; (2) attention_fwd4 with fewer idle states in front of a block's first softmax step (s_nop 5 / 2 / 0; the
# hazard distance stays >= 12 in the generated code: 18 / 15 / 13).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r05t_ab.txt
: > $O
run() { env "$@" 2>&1 | grep -v amdgpu.ids >> $O; }
run AB_MODES=1 timeout 200 python tools/ab_attention_bwd.py bwd_new
run AB_MODES=1 FK_LIB_PATH=build_ab/base/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention_bwd.py bwd_base
run AB_MODES=1 timeout 200 python tools/ab_attention_bwd.py bwd_new
run timeout 200 python tools/ab_attention.py ew5
run FK_LIB_PATH=build_ab/a4_ew2/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py ew2
run FK_LIB_PATH=build_ab/a4_ew0/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py ew0
run timeout 200 python tools/ab_attention.py ew5
run FK_LIB_PATH=build_ab/a4_ew2/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py ew2
cat $O
( timeout 900 python -m pytest -x -q -s tests/test_hip_backward.py -k attention tests/test_hip_cfg5.py > gpurun_out/r05t_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05t_tests.log ); grep -E "passed|failed|Error" gpurun_out/r05t_tests.log | tail -4
for v in ew2 ew0; do ( FK_LIB_PATH=$PWD/build_ab/a4_$v/gpt_image_edit_amd/libfk_gfx950.so timeout 600 python -m pytest -x -q tests/test_hip_kernels.py -k attention > gpurun_out/r05t_tests_$v.log 2>&1; echo "pytest $v rc=$?" | tee -a gpurun_out/r05t_tests_$v.log ); tail -2 gpurun_out/r05t_tests_$v.log; done
