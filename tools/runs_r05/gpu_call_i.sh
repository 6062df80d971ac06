# Round 5, call I: (1) attention_fwd4 with the softmax started one slot later (13 idle states -> 6 behind each S^T chain), the
# ring barrier in the middle of a tile / at its start, 3 / 4 stages; (2) LDS-DMA requests as opaque statements (no compiler
# vmcnt(0) inside the tile loops) against the builtin, 8-wave forward and the backward passes; checksums; parity tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r05i_attention_ab.txt
L=build_ab
run() { env "$@" 2>&1 | grep -v amdgpu.ids >> $O; }
: > $O
run FK_ATTN_KERNEL=4 timeout 200 python tools/ab_attention.py k4_mid_s3
run FK_ATTN_KERNEL=4 FK_LIB_PATH=$L/a4_m0/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py k4_start
run FK_ATTN_KERNEL=4 FK_LIB_PATH=$L/a4_m1s4/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py k4_mid_s4
run FK_ATTN_KERNEL=8 timeout 200 python tools/ab_attention.py k8_opaque
run FK_ATTN_KERNEL=8 FK_LIB_PATH=$L/f8_builtin/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py k8_builtin
run FK_ATTN_KERNEL=4 timeout 200 python tools/ab_attention.py k4_mid_s3
run timeout 200 python tools/ab_attention_bwd.py bwd_opaque
run FK_LIB_PATH=$L/bwd_builtin/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention_bwd.py bwd_builtin
run timeout 200 python tools/ab_attention_bwd.py bwd_opaque
cat $O
( FK_ATTN_KERNEL=4 timeout 900 python -m pytest -x -q tests/test_hip_kernels.py -k attention tests/test_hip_cfg3.py -k "attention or stream" > gpurun_out/r05i_tests_attn4.log 2>&1; echo "pytest attn4 rc=$?" | tee -a gpurun_out/r05i_tests_attn4.log ); tail -3 gpurun_out/r05i_tests_attn4.log
( timeout 900 python -m pytest -x -q tests/test_hip_kernels.py -k attention tests/test_hip_training.py -k attention > gpurun_out/r05i_tests_attn8.log 2>&1; echo "pytest attn8+bwd rc=$?" | tee -a gpurun_out/r05i_tests_attn8.log ); tail -3 gpurun_out/r05i_tests_attn8.log
