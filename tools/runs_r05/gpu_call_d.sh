# Round 5, call D: (1) power probes: 16x16x32 + reads + DMA, DMA cache policies, the 4-wave main-loop skeleton; (2) the 4-wave x
# 64-row attention forward (FK_ATTN_KERNEL=4) against the 8-wave kernel: rates + checksums (must be bit-identical), in the
# shipped build and in a -fno-slp-vectorize build; its parity tests; (3) GPU tests of the per-call launch-control refactor.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 120 tools/build/power_probe 1.2 > gpurun_out/r05d_power_probe.txt 2>&1; echo "probe rc=$?" ); tail -9 gpurun_out/r05d_power_probe.txt
V=$PWD/build_ab/attn_noslp/gpt_image_edit_amd/libfk_gfx950.so
( FK_ATTN_KERNEL=8 timeout 200 python tools/ab_attention.py k8 ; FK_ATTN_KERNEL=4 timeout 200 python tools/ab_attention.py k4 ; FK_LIB_PATH=$V FK_ATTN_KERNEL=4 timeout 200 python tools/ab_attention.py k4_noslp ; FK_LIB_PATH=$V FK_ATTN_KERNEL=8 timeout 200 python tools/ab_attention.py k8_noslp ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05d_attention_4wave_ab.txt; cat gpurun_out/r05d_attention_4wave_ab.txt
( FK_ATTN_KERNEL=4 timeout 600 python -m pytest -x -q tests/test_hip_kernels.py -k attention tests/test_hip_cfg3.py -k "attention or stream" > gpurun_out/r05d_tests_attn4.log 2>&1; echo "pytest attn4 rc=$?" | tee -a gpurun_out/r05d_tests_attn4.log ); tail -3 gpurun_out/r05d_tests_attn4.log
( timeout 900 python -m pytest -q tests/test_hip_kernels.py tests/test_hip_backward.py tests/test_hip_cfg3.py tests/test_hip_train_step.py tests/test_hip_pipeline.py -k "not 28_step" > gpurun_out/r05d_tests_refactor.log 2>&1; echo "pytest refactor rc=$?" | tee -a gpurun_out/r05d_tests_refactor.log ); grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r05d_tests_refactor.log | tail
