# Round 5, call W (last GPU seconds of the round): the K-major GEMM forms (layouts 1 / 2, the weight gradients of the train step)
# with their LDS-DMA requests as opaque statements -- no compiler vmcnt(0) in their K loops -- against the previous build; tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r05w_gemm_kmajor_ab.txt
( echo "== opaque requests"; timeout 100 python tools/ab_gemm_layouts.py; echo "== builtin requests (previous build)"; FK_LIB_PATH=build_ab/before_kmajor/gpt_image_edit_amd/libfk_gfx950.so timeout 100 python tools/ab_gemm_layouts.py ) 2>&1 | grep -v amdgpu.ids > $O; cat $O
( timeout 200 python -m pytest -x -q tests/test_hip_kernels.py tests/test_hip_train_step.py -k "k_major" > gpurun_out/r05w_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05w_tests.log ); tail -2 gpurun_out/r05w_tests.log
