# Round 5, call O: attention_fwd4 with every tile body compiled for its ring stage (no address arithmetic in the loop) against
# the previous build; parity tests incl. the two-kernel bit-equality test.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r05o_attention_ab.txt
: > $O
run() { env "$@" 2>&1 | grep -v amdgpu.ids >> $O; }
run timeout 200 python tools/ab_attention.py static_stage
run FK_LIB_PATH=build_ab/base/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py callJ_d2e1
run timeout 200 python tools/ab_attention.py static_stage
cat $O
( timeout 900 python -m pytest -x -q tests/test_hip_kernels.py tests/test_hip_cfg3.py tests/test_hip_training.py -k "attention or stream" > gpurun_out/r05o_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05o_tests.log ); tail -3 gpurun_out/r05o_tests.log
