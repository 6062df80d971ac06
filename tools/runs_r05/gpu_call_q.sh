# Round 5, call Q: attention backward with the elementwise stage as two in-place sweeps (exponentials, then multiplies) against
# the previous build: ms per call, checksums (bit-identical expected), the backward / training tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r05q_attention_bwd_ab.txt
: > $O
run() { env "$@" 2>&1 | grep -v amdgpu.ids >> $O; }
run timeout 200 python tools/ab_attention_bwd.py two_sweeps
run FK_LIB_PATH=build_ab/base/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention_bwd.py before
run timeout 200 python tools/ab_attention_bwd.py two_sweeps
run FK_LIB_PATH=build_ab/base/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention_bwd.py before
cat $O
( timeout 900 python -m pytest -x -q tests/test_hip_backward.py tests/test_hip_training.py tests/test_hip_train_step.py tests/test_hip_cfg5.py > gpurun_out/r05q_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05q_tests.log ); tail -3 gpurun_out/r05q_tests.log
